"""rocprofv3 --pmc passes over bench.py (scripts/r06_gemm_traffic.sh) -> fabric-side traffic of one training step BY KERNEL:
FETCH_SIZE (x 2: the gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE, L2 hit rate, DRAM credit stalls per
fabric read request.  Steps are delimited by the multi-tensor AdamW launches; the first (warm-up) step is excluded.
usage: r06_traffic_table.py <dir with the pass sub-directories>"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
launches, dur = defaultdict(float), defaultdict(list)
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    disp = {}
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), (int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["End_Timestamp"])))
    order = sorted(disp, key=lambda d: disp[d][0])
    marks = [i for i, d in enumerate(order) if "adamw_multi_kernel" in disp[d][1]]
    if len(marks) < 2:
        continue
    n = len(marks) - 1
    window = set(order[marks[0] + 1: marks[-1] + 1])
    counters = {r["Counter_Name"] for r in rows}
    for r in rows:
        if int(r["Dispatch_Id"]) in window:
            acc[r["Kernel_Name"][:84]][r["Counter_Name"]] += float(r["Counter_Value"]) / n
    if "FETCH_SIZE" in counters:
        for d in window:
            k = disp[d][1][:84]
            launches[k] += 1.0 / n
            dur[k].append(disp[d][2] - disp[d][0])
byt = lambda v: v.get("FETCH_SIZE", 0) * 2048 + v.get("WRITE_SIZE", 0) * 1024
tot = sum(byt(v) for v in acc.values())
gemm = sum(byt(v) for k, v in acc.items() if "gemm" in k)
print(f"# Fabric-side traffic of one training step by kernel — Qwen2-Audio-7B headline, {tot / 1e12:.3f} TB/step "
      f"({gemm / 1e12:.3f} TB in the hand-written GEMMs)\n")
print("FETCH_SIZE / WRITE_SIZE count the L2's memory-side requests: Infinity-Cache hits are INCLUDED (MI355X_MICROARCH.md), "
      "so this is traffic on the fabric, an upper bound of what reaches HBM.\n")
print("| kernel | launches/step | avg us (profiled) | fetch GB/step | fetch MB/launch | write GB/step | L2 hit rate | "
      "fabric read TB/s while it runs | DRAM credit stalls per read request |")
print("|---|---|---|---|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -byt(kv[1]))[:24]:
    fetch, write = v.get("FETCH_SIZE", 0) * 2048, v.get("WRITE_SIZE", 0) * 1024
    hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    us = sum(dur[k]) / max(1, len(dur[k])) / 1e3
    per = fetch / max(launches[k], 1e-9)
    print(f"| `{k}` | {launches[k]:.0f} | {us:.1f} | {fetch / 1e9:.1f} | {per / 1e6:.0f} | {write / 1e9:.1f} | "
          f"{hit / max(1.0, hit + miss):.2f} | {per / max(us, 1e-9) / 1e6:.2f} | "
          f"{v.get('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum', 0) / max(1.0, v.get('TCC_EA0_RDREQ_sum', 0)):.2f} |")
