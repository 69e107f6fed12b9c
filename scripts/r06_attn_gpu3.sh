#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/r06_attn_bwd_ab.py > gpurun_out/r06_attn_bwd_ab.log 2>&1; echo "rc=$?" >> gpurun_out/r06_attn_bwd_ab.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or packed_mask or config_d" > gpurun_out/r06_attn_tests_dq.log 2>&1; echo "rc=$?" >> gpurun_out/r06_attn_tests_dq.log
for s in 0 2; do TN_ATTN_FWD_SCHEDULE=$s TN_ATTN_BWD_DQ=0 timeout 600 python -m pytest "tests/test_parallel_gpu.py::test_flat_data_parallel_engine_over_a_single_rank_rccl_group_trains_like_the_plain_model" -q -x 2>&1 | grep -E "AssertionError: \(|passed|failed" > gpurun_out/r06_flat_engine_sched$s.log; done
tail -4 gpurun_out/r06_attn_tests_dq.log; cat gpurun_out/r06_attn_bwd_ab.log gpurun_out/r06_flat_engine_sched*.log
