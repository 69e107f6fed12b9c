#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_fused_gpu.py -q -x -m gpu -k "rotary" 2>&1 | tail -2
bash scripts/r06_gpu10.sh
