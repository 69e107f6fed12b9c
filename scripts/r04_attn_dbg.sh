#!/bin/bash
out=gpurun_out/${1:-r04f}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python scripts/r04_attn_cmp.py 2>&1 | grep -v amdgpu.ids | tee $out/cmp.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -k "attention" -q 2>&1 | tail -15 | tee $out/pytest_tail.log
