"""Print the kernel timeline (queue, start, duration, gap to the previous kernel's end) around a few RCCL kernels of a
rocprofv3 kernel trace — what ran before / beside / behind a reduce-scatter of the flat engine."""
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "oneRank" in r["Kernel_Name"]]
t0 = int(rows[0]["Start_Timestamp"])
for k in idx[len(idx) // 2: len(idx) // 2 + 3]:
    print("----")
    for r in rows[max(0, k - 6): k + 7]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"q{r.get('Queue_Id', '?'):>3} st{r.get('Stream_Id', '?'):>3} start {(s - t0) / 1e3:12.1f} us  dur {(e - s) / 1e3:9.1f} us  "
              f"grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} lds {r.get('LDS_Block_Size', '?'):>7} {r['Kernel_Name'][:70]}")
