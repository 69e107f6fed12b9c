#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r04d; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --workload qwen2_audio_7b_long --cp 4 --emulate-rank 0 --emulate-shards 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
python scripts/summarize_rocprof.py $f $out/config_D_rank0_kernel_stats.md > /dev/null && head -30 $out/config_D_rank0_kernel_stats.md | cut -c1-200
rm -rf $out/prof
