#!/bin/bash
# Round 6: where the step's fabric-side traffic comes from, per kernel — FETCH_SIZE / WRITE_SIZE (the
# step_traffic.sh passes, which also refresh roofline.traffic), L2 hits / misses, fabric read requests and DRAM credit stalls.
#   usage (GPU box, repo root): bash scripts/r06_gemm_traffic.sh  -> gpurun_out/r06_step_traffic_by_kernel.md
R=$(pwd); export TMPDIR=/tmp
rm -rf gpurun_out/step_traffic
TN_ROUND=r06 bash scripts/step_traffic.sh qwen2_audio_7b > gpurun_out/r06_step_traffic.log 2>&1
cd /tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/step_traffic/$tag --output-format csv -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $R/gpurun_out/step_traffic_$tag.log 2>&1
done
cd $R
python scripts/r06_traffic_table.py gpurun_out/step_traffic > gpurun_out/r06_step_traffic_by_kernel.md
cat gpurun_out/r06_step_traffic_by_kernel.md; tail -2 gpurun_out/r06_step_traffic.log | cut -c1-400
rm -rf gpurun_out/step_traffic
