#!/bin/bash
# configs D and E on ONE MI355X: tests of the parallel paths, then emulated-rank bench lines (bench.py --emulate-rank)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q > gpurun_out/r03h_parallel_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/r03h_parallel_tests.log
tail -5 gpurun_out/r03h_parallel_tests.log
for r in 0 3; do
  timeout 900 python bench.py --workload qwen2_audio_7b_long --cp 4 --emulate-rank $r --steps 4 --warmup 2 --no-cpu-baseline \
      > gpurun_out/r03h_bench_D_cp4_rank$r.json 2> gpurun_out/r03h_bench_D_cp4_rank$r.err; echo "D rank $r exit $?"; tail -c 600 gpurun_out/r03h_bench_D_cp4_rank$r.json; tail -3 gpurun_out/r03h_bench_D_cp4_rank$r.err
done
for r in 0; do
  timeout 900 python bench.py --workload kimi_audio_7b --tp 2 --emulate-rank $r --steps 4 --warmup 2 --no-cpu-baseline \
      > gpurun_out/r03h_bench_E_tp2_rank$r.json 2> gpurun_out/r03h_bench_E_tp2_rank$r.err; echo "E rank $r exit $?"; tail -c 600 gpurun_out/r03h_bench_E_tp2_rank$r.json; tail -3 gpurun_out/r03h_bench_E_tp2_rank$r.err
done
