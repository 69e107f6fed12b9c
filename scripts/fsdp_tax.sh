#!/bin/bash
# Data parallelism on ONE GPU over a 1-rank RCCL mesh (TN_FORCE_FSDP=1) vs the plain model: what each engine adds to the step
# on every rank (copies, casts, the sharded optimizer, hooks).  flat = utils/zero_dp.py, fsdp2 = torch fully_shard.
#   usage (GPU box, repo root): bash scripts/fsdp_tax.sh [--prof]
R=$(pwd)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
line() { grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms,', d['peak_mem_GB_rank0'], 'GB,', d['config']['parallelism'])"; }
echo "== plain";                 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | line
echo "== flat engine, 1 rank";   TN_FORCE_FSDP=1 python bench.py --dp-engine flat --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-rooflines 2>gpurun_out/tax_flat.err | line
echo "== FSDP2, 1 rank";         TN_FORCE_FSDP=1 python bench.py --dp-engine fsdp2 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-rooflines 2>gpurun_out/tax_fsdp2.err | line
echo "== plain (again)";         python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | line
[ "$1" = "--prof" ] || exit 0
cd /tmp; export TMPDIR=/tmp
TN_FORCE_FSDP=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/flat_prof --output-format csv -- python $R/bench.py --dp-engine flat --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $R/gpurun_out/flat_prof.log 2>&1
cd $R
python scripts/summarize_rocprof.py gpurun_out/flat_prof > gpurun_out/flat_prof_summary.md 2>/dev/null || true
