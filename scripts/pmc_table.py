"""rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` counter_collection CSVs -> markdown rows.
usage: pmc_table.py <fetch.csv> <write.csv> name=read_MB:write_MB ...   (algorithmic MB per launch, per kernel-name
substring).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section): the read column is doubled."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return acc


rd, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("| kernel | algorithmic read MB | FETCH_SIZE x2 MB | algorithmic write MB | WRITE_SIZE MB | avg µs (profiled) | algorithmic GB/s |")
print("|---|---|---|---|---|---|---|")
for spec in sys.argv[3:]:
    name, v = spec.split("=")
    a_rd, a_wr = [float(t) for t in v.split(":")]
    kr = [k for k in rd if name in k]
    if not kr:
        print(f"| {name} | (not found) |")
        continue
    k = kr[0]
    f = sum(c for c, _ in rd[k]) / len(rd[k]) * 1024 / 1e6 * 2
    w = sum(c for c, _ in wr[k]) / len(wr[k]) * 1024 / 1e6 if k in wr else float("nan")
    us = sum(t for _, t in rd[k]) / len(rd[k]) / 1e3
    print(f"| {name} | {a_rd:.0f} | {f:.0f} | {a_wr:.0f} | {w:.0f} | {us:.1f} | {(a_rd + a_wr) * 1e6 / (us * 1e-6) / 1e9:.0f} |")
