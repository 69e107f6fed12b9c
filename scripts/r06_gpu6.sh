#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gemm_fused_gpu.py tests/test_reference_fixtures_gpu.py tests/test_full_size_parity_gpu.py -q -x -m gpu > gpurun_out/r06_fusion_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_fusion_tests.log
tail -6 gpurun_out/r06_fusion_tests.log
for rep in 1 2; do
  for sw in "1 1" "0 0"; do
    set -- $sw
    TN_RESIDUAL_IN_EPILOGUE=$1 TN_GELU_EPILOGUE=$2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('residual/gelu epilogues $1/$2:', d['ms_per_step'], 'ms  loss', d['loss_per_sample_last'])" | tee -a gpurun_out/r06_fusion_ab.log
  done
done
