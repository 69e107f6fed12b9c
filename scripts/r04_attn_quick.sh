#!/bin/bash
# Round 4: attention parity tests + the two attention benches on the library as built (fused dK+dV by default).
out=gpurun_out/${1:-r04q}; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -k "attention" -x -q > $out/pytest_attention.log 2>&1
echo "pytest rc=$?" | tee -a $out/summary.log
tail -3 $out/pytest_attention.log | tee -a $out/summary.log
timeout 600 python scripts/attn_sched_bench.py 2>&1 | grep -v "amdgpu.ids" | tee -a $out/summary.log
timeout 600 python scripts/attn_long_bench.py 2>&1 | grep -v "amdgpu.ids" | tee -a $out/summary.log
