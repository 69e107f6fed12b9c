"""GPU idle time inside a training step from a rocprofv3 --kernel-trace CSV (no PMC: counter collection serialises
launches): span of the kernels between consecutive optimizer steps vs the sum of their durations.
    python scripts/step_gaps.py <dir-with-kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"]]
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1:b + 1]
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    busy_union, prev_end, gaps = 0, int(seg[0]["Start_Timestamp"]), []
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > prev_end:
            gaps.append((s - prev_end, r["Kernel_Name"][:60]))
        busy_union += max(0, e - max(s, prev_end))
        prev_end = max(prev_end, e)
    gaps.sort(reverse=True)
    print(f"step: {len(seg)} kernels, span {span / 1e6:.1f} ms, GPU busy {busy_union / 1e6:.1f} ms, idle "
          f"{(span - busy_union) / 1e6:.1f} ms ({len(gaps)} gaps, median {gaps[len(gaps) // 2][0] / 1e3:.1f} us)")
    print("  largest gaps (us, next kernel):", [(round(g / 1e3), n) for g, n in gaps[:6]])
