#!/bin/bash
# round 6, final evidence on one box: HBM traffic of the step (refreshes roofline.traffic), the driver's default bench line with
# the CPU baseline, the kernel statistics of the same command, the whole GPU suite, smoke()
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/step_traffic
TN_ROUND=r06 bash scripts/step_traffic.sh qwen2_audio_7b > gpurun_out/r06_step_traffic.log 2>&1
rm -rf gpurun_out/step_traffic
cp gpurun_out/r06_step_hbm_traffic_qwen2_audio_7b.json profiles/ 2>/dev/null      # (so that the bench line below finds it)
python bench.py > gpurun_out/r06z_bench_default.json 2> gpurun_out/r06z_bench_default.err
tail -c 600 gpurun_out/r06z_bench_default.json
bash scripts/r06_step_measure.sh r06z > gpurun_out/r06z_measure.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06z_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r06z_smoke.log
bash scripts/r06_gpu_suite.sh
