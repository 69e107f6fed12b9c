#!/bin/bash
# which hardware queue do the RCCL kernels of the flat engine land on, and do they run beside compute?  (one short profiled run
# per setting; prints the queue ids of GEMM / RCCL kernels and the overlap summary)
R=$(pwd); out=$R/gpurun_out/r05q; mkdir -p $out; export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
B="--dp-engine flat --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-rooflines"
run() {  # name, env...
  name=$1; shift
  cd /tmp
  env TN_FORCE_FSDP=1 TN_DP_FORCE_COLLECTIVES=1 "$@" rocprofv3 --kernel-trace -d $out/$name --output-format csv -- python $R/bench.py $B > $out/$name.log 2>&1
  cd $R
  python - $out/$name $name <<'PY'
import csv, glob, sys, collections
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
q = collections.defaultdict(collections.Counter)
for r in rows:
    kind = "rccl" if "oneRank" in r["Kernel_Name"] or "nccl" in r["Kernel_Name"].lower() else "gemm" if "tn::gemm" in r["Kernel_Name"] else "other"
    q[kind][(r["Queue_Id"], r["Stream_Id"])] += 1
print(sys.argv[2], {k: dict(v) for k, v in q.items() if k != "other"})
PY
  mkdir -p $out/cmp/$name; mv $out/$name/* $out/cmp/$name/ 2>/dev/null
}
run prio_default TN_COMM_STREAM_PRIORITY=0 TORCH_NCCL_HIGH_PRIORITY=0
run prio_high TN_X=1
run prio_high_per_tile TN_GEMM_PERSIST=0
run hwq8 TN_COMM_STREAM_PRIORITY=0 TORCH_NCCL_HIGH_PRIORITY=0 GPU_MAX_HW_QUEUES=8
for n in prio_default prio_high prio_high_per_tile hwq8; do
  python - $out/cmp/$n $n <<'PY'
import sys
sys.path.insert(0, "scripts")
import r05_overlap_timeline as T
rows = T.load(sys.argv[1])
sel, wall, n = T.steady(rows)
isc = lambda x: "oneRank" in x or "nccl" in x.lower()
cu = T.union([(r["s"], r["e"]) for r in sel if isc(r["name"])])
pu = T.union([(r["s"], r["e"]) for r in sel if not isc(r["name"])])
busy = sum(e - s for s, e in cu); hid = T.overlap(cu, pu)
print(f"{sys.argv[2]:22s} wall {wall:7.1f} ms/step  RCCL busy {busy / n / 1e6:6.2f} ms  beside compute {hid / max(busy, 1) * 100:4.0f} %  comm-only {(busy - hid) / n / 1e6:6.2f} ms")
PY
done
rm -rf $out/cmp
