#!/bin/bash
# (the TN_GEMM_TILE_ORDER variant this script built and timed is recorded in profiles/r04f_gemm_tile_order_xcd_round_robin_neutral.log;
#  the macro is not in csrc/gemm.hip: measured neutral, not kept)
out=gpurun_out/r04t; mkdir -p $out
V=$(pwd)/touchnet_amd/_lib/variants
timeout 600 env TN_AMD_LIB=$V/order1/libtouchnet_amd.so python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -2
run() {
  TN_AMD_LIB=$2 timeout 900 python bench.py --steps 8 --warmup 3 --no-kernel-rooflines --no-cpu-baseline > $out/b.json 2> $out/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/b.json").read().strip().splitlines()[-1])
    print("$1", d["ms_per_step"], "ms", d["value"], "tok/s loss", d.get("loss_per_sample_last"))
except Exception as e:
    print("$1 failed", e); print(open("$out/b.err").read()[-1500:])
PY
}
{
run default ""
run order1 $V/order1/libtouchnet_amd.so
run default ""
run order1 $V/order1/libtouchnet_amd.so
} 2>&1 | tee $out/summary.log
