#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r04k; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --workload kimi_audio_7b_speech --tp 2 --emulate-rank 0 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
python scripts/summarize_rocprof.py $f $out/kimi_speech_kernel_stats.md > /dev/null && head -45 $out/kimi_speech_kernel_stats.md | cut -c1-200
rm -rf $out/prof
