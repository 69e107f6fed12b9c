#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_fused_gpu.py -q -x -m gpu > gpurun_out/r06_fusion_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_fusion_tests.log
tail -3 gpurun_out/r06_fusion_tests.log
: > gpurun_out/r06_fusion_ab2.log
for rep in 1 2; do
  for sw in "0 1" "0 0" "1 1"; do
    set -- $sw
    TN_RESIDUAL_IN_EPILOGUE=$1 TN_GELU_EPILOGUE=$2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('residual epilogue $1 / gelu epilogue $2:', d['ms_per_step'], 'ms  loss', d['loss_per_sample_last'])" | tee -a gpurun_out/r06_fusion_ab2.log
  done
done
