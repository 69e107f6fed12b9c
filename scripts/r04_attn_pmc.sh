#!/bin/bash
# Round 4: kernel times + PMC of the attention backward kernels (fused dK+dV, dQ) at a config-D-like length and on the
# headline documents.  usage: bash scripts/r04_attn_pmc.sh <tag>   -> gpurun_out/<tag>/
R=$(pwd); out=$R/gpurun_out/${1:-r04b}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for wl in long docs; do
  rm -rf $out/trace_$wl
  rocprofv3 --kernel-trace --stats -d $out/trace_$wl --output-format csv -- python $R/scripts/attn_prof.py $wl > /dev/null 2>&1
  f=$(find $out/trace_$wl -name "*kernel_stats.csv" | head -1); echo "== $wl kernel stats" >> $out/summary.log; head -8 $f | cut -c1-200 >> $out/summary.log
  cp $f $out/kernel_stats_$wl.csv; rm -rf $out/trace_$wl
  rm -rf $R/gpurun_out/attn_pmc
  (cd $R; bash scripts/attn_pmc.sh $wl > /dev/null 2>&1; cp gpurun_out/attn_pmc_final.md $out/pmc_$wl.md; rm -rf gpurun_out/attn_pmc)
done
cat $out/summary.log; cat $out/pmc_long.md
