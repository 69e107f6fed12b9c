#!/bin/bash
# Round 5 final numbers on one MI355X: the driver's default bench line (kernel rooflines + cpu_baseline), the kernel statistics
# of the step (rocprofv3 --kernel-trace --stats of the same command), whole-step HBM traffic (PMC, separate passes), then the
# bench line again so that roofline.traffic is picked up with a matching source digest.   usage: bash scripts/r05_final_measure.sh <tag>
R=$(pwd); tag=${1:-r05z}; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['step_mfu'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('dominant_kernel',{}).get('kernel'))" | tee -a $out/summary.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
cd $R
f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
python scripts/summarize_rocprof.py $f $out/qwen2audio7b_kernel_stats.md > /dev/null && head -16 $out/qwen2audio7b_kernel_stats.md | tee -a $out/summary.log
rm -rf $out/prof
TN_ROUND=r05 bash scripts/step_traffic.sh qwen2_audio_7b > $out/traffic.log 2>&1; tail -2 $out/traffic.log | cut -c1-600 | tee -a $out/summary.log
cp gpurun_out/r05_step_hbm_traffic_qwen2_audio_7b.json $out/ 2>/dev/null; rm -rf gpurun_out/step_traffic
mkdir -p profiles; cp gpurun_out/r05_step_hbm_traffic_qwen2_audio_7b.json profiles/ 2>/dev/null
python bench.py --no-cpu-baseline > $out/bench_with_traffic.json 2> $out/bench_with_traffic.err
python3 -c "
import json; d=json.loads(open('$out/bench_with_traffic.json').read().strip().splitlines()[-1]); print('bench+traffic', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_source','')[:60])" | tee -a $out/summary.log
