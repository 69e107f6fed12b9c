#!/bin/bash
# rocprofv3 kernel traces of the plain step and of the flat engine on a 1-rank RCCL mesh; per-kernel-name time per step of both.
R=$(pwd)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/plain_prof --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $R/gpurun_out/plain_prof.log 2>&1
TN_FORCE_FSDP=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/flat_prof --output-format csv -- python $R/bench.py --dp-engine ${1:-flat} --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $R/gpurun_out/flat_prof.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
def per_step(d):
    f = glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"]]
    # steps are separated by the LAST adamw launch of each step: group consecutive marks
    ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 50]
    n = len(ends) - 1
    acc = collections.defaultdict(lambda: [0, 0])
    for r in rows[ends[0] + 1: ends[-1] + 1]:
        a = acc[r["Kernel_Name"][:100]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = (int(rows[ends[-1]]["End_Timestamp"]) - int(rows[ends[0]]["End_Timestamp"])) / n / 1e6
    return {k: (c / n, t / n / 1e6) for k, (c, t) in acc.items()}, span
a, sa = per_step("plain_prof")
b, sb = per_step("flat_prof")
print(f"wall per step: plain {sa:.1f} ms, engine {sb:.1f} ms; kernel time {sum(v[1] for v in a.values()):.1f} vs {sum(v[1] for v in b.values()):.1f}")
keys = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
for k in keys[:40]:
    pa, pb = a.get(k, (0, 0)), b.get(k, (0, 0))
    print(f"{pb[1] - pa[1]:+8.2f} ms  plain {pa[1]:8.2f} ({pa[0]:6.0f}x)  engine {pb[1]:8.2f} ({pb[0]:6.0f}x)  {k}")
PY
