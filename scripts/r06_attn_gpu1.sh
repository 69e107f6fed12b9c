#!/bin/bash
# round 6: stream forward A/B + the attention parity tests with the stream schedule forced + the wave trace
mkdir -p gpurun_out
HPWS=${HPWS:-1,2,4} timeout 600 python scripts/r06_attn_ab.py > gpurun_out/r06_attn_ab.log 2>&1; echo "ab rc=$?" >> gpurun_out/r06_attn_ab.log
TN_ATTN_FWD_SCHEDULE=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or packed_mask or config_d" > gpurun_out/r06_attn_tests_sched2.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06_attn_tests_sched2.log
HPW=1 timeout 300 python scripts/r06_attn_trace.py > gpurun_out/r06_attn_trace_hpw1.log 2>&1
tail -5 gpurun_out/r06_attn_tests_sched2.log; cat gpurun_out/r06_attn_ab.log; cat gpurun_out/r06_attn_trace_hpw1.log
