#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_fused_gpu.py -q -x -m gpu > gpurun_out/r06_fusion_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_fusion_tests.log
tail -4 gpurun_out/r06_fusion_tests.log
R=$(pwd)
for sw in "1 1" "0 0"; do
  set -- $sw
  cd /tmp
  TN_RESIDUAL_IN_EPILOGUE=$1 TN_GELU_EPILOGUE=$2 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$1$2 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > /dev/null 2>&1
  cd $R
  f=$(ls gpurun_out/prof_$1$2/*/*kernel_trace.csv | head -1)
  python scripts/summarize_rocprof.py $f gpurun_out/r06_kernel_stats_fusions_$1$2.md > /dev/null
  rm -rf gpurun_out/prof_$1$2
  echo "== epilogues $1 $2"; head -44 gpurun_out/r06_kernel_stats_fusions_$1$2.md | cut -c1-150
done
