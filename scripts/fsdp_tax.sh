#!/bin/bash
# FSDP2 over a 1-rank RCCL mesh vs the unsharded model on ONE GPU: step time of both + rocprofv3 kernel stats of the
# sharded run (what FSDP2's copy-in/out, casts and the sharded optimizer add on every rank at N = 8).
#   usage (GPU box, repo root): bash scripts/fsdp_tax.sh
R=$(pwd)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
echo "== unsharded"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['peak_mem_GB_rank0'])"
echo "== FSDP2, 1-rank RCCL mesh"; TN_FORCE_FSDP=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-rooflines 2>gpurun_out/fsdp_force.err | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['peak_mem_GB_rank0'])"
[ "$1" = "--no-prof" ] && exit 0
cd /tmp; export TMPDIR=/tmp
TN_FORCE_FSDP=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fsdp_prof --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $R/gpurun_out/fsdp_prof.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/fsdp_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"]]
n = len(marks) - 1
acc = collections.defaultdict(lambda: [0, 0])
for r in rows[marks[0] + 1: marks[-1] + 1]:
    a = acc[r["Kernel_Name"][:90]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in acc.values()) / n / 1e6
print(f"kernel time per step {tot:.1f} ms over {n} steps")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t / n / 1e6:9.2f} ms {c / n:8.1f} x  {k}")
PY
