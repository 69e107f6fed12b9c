#!/bin/bash
# round 6: the default bench line (with cpu_baseline) and the kernel statistics of the same command, final sources
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py > gpurun_out/r06z_bench_default.json 2> gpurun_out/r06z_bench_default.err
bash scripts/r06_step_measure.sh r06z > gpurun_out/r06z_measure.log 2>&1
cat gpurun_out/r06z/summary.log | head -3
