#!/bin/bash
root=$(pwd); lib=$root/touchnet_amd/_lib; out=$root/gpurun_out/r04x; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for md in 790 100 8192; do
for v in 0 64; do
  rm -rf $out/tr
  TN_AMD_LIB=$lib/variants/fkv$v/libtouchnet_amd.so timeout 300 rocprofv3 --kernel-trace --stats -d $out/tr --output-format csv -- python $root/scripts/r04_fkv_fixed.py $md > /dev/null 2>&1
  f=$(find $out/tr -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" "$md" <<'PY' | tee -a $out/fixed.log
import csv, sys
rows = {r["Name"].split("(")[0].split("::")[-1]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "attn_" in r["Name"]}
print("docs~%s ABL=%s " % (sys.argv[3], sys.argv[2]) + "  ".join("%s %.0f us" % (k.split("<")[0], v) for k, v in sorted(rows.items())))
PY
done; done; rm -rf $out/tr
