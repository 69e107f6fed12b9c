#!/bin/bash
# Per-kernel mean durations of the attention kernels on one workload of scripts/attn_prof.py, for the default library
# and for variant builds (scripts/build_variant.sh): rocprofv3 kernel trace, one run per library.
# usage: scripts/attn_ktimes.sh <docs|causal|tower> [variant ...]      ("default" = the in-tree library)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
wl=$1; shift
for v in "$@"; do
  out=$root/gpurun_out/ktimes_${wl}_$v
  rm -rf "$out"
  if [ "$v" = default ]; then unset TN_AMD_LIB; else export TN_AMD_LIB=$root/touchnet_amd/_lib/variants/$v/libtouchnet_amd.so; fi
  rocprofv3 --kernel-trace --stats -d "$out" --output-format csv -- python "$root/scripts/attn_prof.py" "$wl" > /dev/null 2>&1
  echo "== $wl / $v"
  python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    n = row["Name"]
    if "attn_" in n:
        print(f"  {n.split('(')[0][:60]:60s} calls {row['Calls']:>4s}  avg {float(row['AverageNs']) / 1e3:9.1f} us")
PY
done
