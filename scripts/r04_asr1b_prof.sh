#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r04b1; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --workload llama_asr_1b --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
python scripts/summarize_rocprof.py $f $out/llama_asr_1b_kernel_stats.md > /dev/null && head -28 $out/llama_asr_1b_kernel_stats.md | cut -c1-170
rm -rf $out/prof
