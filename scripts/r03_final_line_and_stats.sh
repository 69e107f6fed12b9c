R=$(pwd); mkdir -p gpurun_out
python bench.py > gpurun_out/r03v_bench_default.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r03v_bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_mfu'], d['roofline']['frac'], d['roofline']['traffic'])"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03v_prof --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $R/gpurun_out/r03v_prof.log 2>&1
cd $R
f=$(ls gpurun_out/r03v_prof/*/*kernel_trace.csv | head -1)
python scripts/summarize_rocprof.py $f gpurun_out/r03v_qwen2audio7b_kernel_stats.md > /dev/null && head -14 gpurun_out/r03v_qwen2audio7b_kernel_stats.md
