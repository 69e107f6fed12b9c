#!/bin/bash
# End-of-round evidence on one MI355X: default bench line (+ kernel rooflines, CPU baseline), LlamaForASR-1B line, rocprofv3
# kernel statistics of the default step, whole-step HBM traffic from the PMC counters (scripts/step_traffic.sh).
R=$(pwd); mkdir -p gpurun_out
python bench.py > gpurun_out/r03k_bench_default.json 2> gpurun_out/r03k_bench_default.err; tail -c 300 gpurun_out/r03k_bench_default.json; echo
python bench.py --workload llama_asr_1b --no-cpu-baseline > gpurun_out/r03k_bench_llama_asr_1b.json 2>/dev/null; tail -c 200 gpurun_out/r03k_bench_llama_asr_1b.json; echo
bash scripts/step_traffic.sh qwen2_audio_7b 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03k_prof --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $R/gpurun_out/r03k_prof.log 2>&1
cd $R
f=$(ls gpurun_out/r03k_prof/*/*kernel_trace.csv | head -1)
python scripts/summarize_rocprof.py $f gpurun_out/r03k_qwen2audio7b_kernel_stats.md && head -14 gpurun_out/r03k_qwen2audio7b_kernel_stats.md
