#!/bin/bash
# Round 5, one call on one box: parity of the fused GEMM launches, their microbench, and the whole step A/B
# (TN_MLP_EPILOGUE / TN_GROUPED_WGRAD off = the round-4 composition), interleaved twice.
out=gpurun_out/r05a; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm_fused_gpu.py -q 2>&1 | tail -25 | tee $out/pytest_fused.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gemm_fused_gpu.py 2>&1 | tail -15 | tee $out/pytest_gpu_all.log
timeout 300 python scripts/r05_mlp_fusion_bench.py 2>&1 | tee $out/mlp_fusion_bench.log
run() {
  env $2 timeout 900 python bench.py --steps 8 --warmup 3 --no-kernel-rooflines --no-cpu-baseline > $out/b.json 2> $out/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/b.json").read().strip().splitlines()[-1])
    print("$1", d["ms_per_step"], "ms", d["value"], "tok/s loss", d.get("loss_per_sample_last"), "mem", d.get("peak_mem_GB_rank0"))
except Exception as e:
    print("$1 failed", e); print(open("$out/b.err").read()[-2500:])
PY
}
{
run fused "TN_X=1"
run round4 "TN_MLP_EPILOGUE=0 TN_GROUPED_WGRAD=0"


run fused "TN_X=1"
run round4 "TN_MLP_EPILOGUE=0 TN_GROUPED_WGRAD=0"
} 2>&1 | tee $out/summary.log
