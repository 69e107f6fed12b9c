#!/bin/bash
# round 6: the whole GPU suite (what the driver runs at round end) -> gpurun_out/r06_gpu_suite.log
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_suite.log 2>&1
echo "rc=$?" >> gpurun_out/r06_gpu_suite.log
tail -15 gpurun_out/r06_gpu_suite.log
