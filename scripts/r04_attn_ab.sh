#!/bin/bash
# Round 4: A/B of the attention backward on one MI355X — the two-launch dV / dK scheme (TN_ATTN_BWD_KV=split) against the
# fused dK+dV pass (default) — parity tests first, then the schedule and long-sequence benches.  Output: gpurun_out/$1/
out=gpurun_out/${1:-r04a}; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -k "attention" -x -q > $out/pytest_attention_fused.log 2>&1
echo "pytest fused rc=$?" | tee -a $out/summary.log
tail -3 $out/pytest_attention_fused.log | tee -a $out/summary.log
for mode in split fused; do
  TN_ATTN_BWD_KV=$mode timeout 600 python scripts/attn_sched_bench.py > $out/sched_$mode.log 2>&1
  TN_ATTN_BWD_KV=$mode timeout 600 python scripts/attn_long_bench.py > $out/long_$mode.log 2>&1
  echo "== $mode" | tee -a $out/summary.log
  cat $out/sched_$mode.log $out/long_$mode.log | grep -v Warning | tee -a $out/summary.log
done
