"""rocprofv3 --pmc counter_collection CSV(s) -> per-kernel mean counter values per launch (markdown).
usage: pmc_sum.py <dir-or-csv> [<dir-or-csv> ...] [--match substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict

paths, match = [], None
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match":
        match = args.pop(0)
    else:
        paths.append(a)
files = []
for p in paths:
    files += glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(p) else [p]
vals = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if match and match not in k:
            continue
        vals[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for k in vals for c in vals[k]})
print("| kernel | launches | " + " | ".join(ctrs) + " |")
print("|---|---|" + "---|" * len(ctrs))
for k in sorted(vals):
    n = max(len(v) for v in vals[k].values())
    print(f"| {k} | {n} | " + " | ".join(f"{sum(vals[k][c]) / max(1, len(vals[k][c])):.4g}" if c in vals[k] else "-" for c in ctrs) + " |")
