#!/bin/bash
# round 6: kernel statistics of the step with the rotary gradient inside / outside the attention backward (same box)
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
for sw in 0 1; do
  out=$R/gpurun_out/r06_rope_grad_prof_$sw; mkdir -p $out
  cd /tmp
  TN_ROPE_GRAD_IN_ATTENTION=$sw rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
  cd $R
  f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
  python scripts/summarize_rocprof.py $f $R/gpurun_out/r06_rope_grad_kernel_stats_$sw.md > /dev/null
  rm -rf $out
  grep -E "attn_|rope_apply|category|packed attention|other hand" $R/gpurun_out/r06_rope_grad_kernel_stats_$sw.md | cut -c1-150
done
