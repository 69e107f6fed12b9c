#!/bin/bash
# round 6: working-tree kernels vs the kernels of HEAD (variant library `head`), alternating processes on one box
mkdir -p gpurun_out; : > gpurun_out/r06_interval_ab.log
for rep in 1 2 3; do
  python scripts/r06_attn_times.py new 2>/dev/null | tee -a gpurun_out/r06_interval_ab.log
  TN_AMD_LIB=touchnet_amd/_lib/variants/head/libtouchnet_amd.so python scripts/r06_attn_times.py head 2>/dev/null | tee -a gpurun_out/r06_interval_ab.log
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attention or attn" 2>&1 | tail -3
