#!/bin/bash
# Round 6: the driver's default bench line + the kernel statistics of the step (rocprofv3 --kernel-trace --stats of the same
# command).   usage: bash scripts/r06_step_measure.sh <tag>   -> gpurun_out/<tag>/
R=$(pwd); tag=${1:-r06m}; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('step_mfu'), d['roofline']['frac'])" | tee -a $out/summary.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
python scripts/summarize_rocprof.py $f $out/qwen2audio7b_kernel_stats.md > /dev/null && head -40 $out/qwen2audio7b_kernel_stats.md | tee -a $out/summary.log
rm -rf $out/prof
