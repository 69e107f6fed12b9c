#!/bin/bash
# A/B of the dK/dV kernels: variant library (round-5 kernels) vs the current one, interleaved, same box; then parity tests
mkdir -p gpurun_out; out=gpurun_out/r06_attn_kv_ab.log; : > $out
for rep in 1 2; do
  for lib in variants/kv_r5/libtouchnet_amd.so libtouchnet_amd.so; do
    echo "== $lib (rep $rep)" >> $out
    TN_AMD_LIB=$PWD/touchnet_amd/_lib/$lib timeout 600 python scripts/r06_attn_bwd_ab.py 2>&1 | grep "^B" | sed 's/dq=old [0-9]*us [0-9]*TF  //' >> $out
  done
done
cat $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or packed_mask or config_d" > gpurun_out/r06_attn_tests_kv.log 2>&1; echo "rc=$?" >> gpurun_out/r06_attn_tests_kv.log
tail -4 gpurun_out/r06_attn_tests_kv.log
