#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_full_size_parity_gpu.py tests/test_bench_multirank_gpu.py tests/test_reference_fixtures_gpu.py "tests/test_parallel_gpu.py::test_weight_gradients_on_a_side_stream_train_identically" -q -x -m gpu -s > gpurun_out/r06_new_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_new_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r06_smoke.log
grep -E "FULL-SIZE|FIXTURE PARITY|passed|failed|Error|rc=" gpurun_out/r06_new_tests.log | tail -30; tail -3 gpurun_out/r06_smoke.log
