#!/bin/bash
# round 6: the GELU epilogues with packed arithmetic + hoisted pre-activation loads (working tree) vs the kernels of HEAD (variant
# library `head`): tests, then kernel statistics of the step, alternating on one box
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_fused_gpu.py tests/test_kernels_gpu.py tests/test_reference_fixtures_gpu.py -q -x -m gpu -k "gelu or tower or qwen2_audio or whisper" 2>&1 | tail -2
for v in head new head new; do
  out=$R/gpurun_out/r06_gelu_$v; rm -rf $out; mkdir -p $out
  cd /tmp
  lib=""; [ $v = head ] && lib=$R/touchnet_amd/_lib/variants/head/libtouchnet_amd.so
  TN_AMD_LIB=$lib rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
  cd $R
  f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
  python scripts/summarize_rocprof.py $f $out/stats.md > /dev/null
  echo "$v: $(grep -E 'gemm16_kernel<false, (false|true), 6, [67]>|wall' $out/stats.md | sed 's/(tn::gemm::Params)//' | cut -c1-120 | tr '\n' ' ')"
  rm -rf $out
done
