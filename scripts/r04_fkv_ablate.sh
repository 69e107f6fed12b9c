#!/bin/bash
# Round 4: where does a loop trip of the fused dK+dV kernel spend its time?  Builds variants of csrc/attn_bwd_fused.hip with
# pieces removed (-DTN_FKV_ABL=<bits>: 1 no LDS-DMA, 2 no barrier, 4 no softmax arithmetic, 8 no LDS reads, 16 no MFMAs,
# 32 no lgkmcnt waits; results are garbage) and times the kernel on a causal T = 32768 batch.
#   usage (build box): bash scripts/r04_fkv_ablate.sh build      (GPU box): bash scripts/r04_fkv_ablate.sh run <tag>
root=$(cd "$(dirname "$0")/.." && pwd); lib=$root/touchnet_amd/_lib
VARS="0 1 2 3 4 8 16 32 40 12 44 47"
if [ "$1" = build ]; then
  for v in $VARS; do
    d=$lib/variants/fkv$v; mkdir -p $d
    (hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DTN_FKV_ABL=$v -c $root/touchnet_amd/csrc/attn_bwd_fused.hip -o $d/attn_bwd_fused.o &&
     hipcc -shared -fPIC --offload-arch=gfx950 $(ls $lib/*.o | grep -v attn_bwd_fused.o) $d/attn_bwd_fused.o -o $d/libtouchnet_amd.so && rm $d/attn_bwd_fused.o) &
  done; wait; ls $lib/variants/fkv*/libtouchnet_amd.so | wc -l
else
  out=$root/gpurun_out/${2:-r04h}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
  for v in $VARS; do
    rm -rf $out/tr
    TN_AMD_LIB=$lib/variants/fkv$v/libtouchnet_amd.so timeout 300 rocprofv3 --kernel-trace --stats -d $out/tr --output-format csv -- python $root/scripts/attn_prof.py long > /dev/null 2>&1
    f=$(find $out/tr -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$v" <<'PY' | tee -a $out/ablate.log
import csv, sys
rows = {r["Name"].split("(")[0].split("::")[-1]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "attn_bwd" in r["Name"]}
print("ABL=%s " % sys.argv[2] + "  ".join("%s %.0f us" % (k.split("<")[0], v) for k, v in sorted(rows.items())))
PY
  done; rm -rf $out/tr
fi
