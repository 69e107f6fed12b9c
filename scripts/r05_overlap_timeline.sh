#!/bin/bash
# Round 5: a REAL overlap timeline of the flat data-parallel engine on one MI355X.  A 1-rank RCCL group
# (TN_FORCE_FSDP=1) with the engine's identity shortcut switched off (TN_DP_FORCE_COLLECTIVES=1): every block's
# reduce-scatter and all-gather is an RCCL kernel on the communication stream, launched where a rank of an 8-GPU job
# launches it.  rocprofv3 --kernel-trace of the step with persistent and with per-tile GEMM launches; for every RCCL
# kernel: how much of it ran beside compute kernels, how long compute was stalled around it, and the step tax against the
# plain step on the same box.   usage: bash scripts/r05_overlap_timeline.sh [tag]
R=$(pwd); tag=${1:-r05o}; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
B="--steps 4 --warmup 2 --no-cpu-baseline --no-kernel-rooflines"
cd /tmp
env -u WORLD_SIZE -u RANK -u LOCAL_RANK rocprofv3 --kernel-trace -d $out/plain --output-format csv -- python $R/bench.py $B > $out/plain.log 2>&1
for mode in persistent per_tile; do
  p=1; [ $mode = per_tile ] && p=0
  TN_GEMM_PERSIST=$p TN_FORCE_FSDP=1 TN_DP_FORCE_COLLECTIVES=1 rocprofv3 --kernel-trace -d $out/$mode --output-format csv -- \
    python $R/bench.py --dp-engine flat $B > $out/$mode.log 2>&1
done
cd $R
python scripts/r05_overlap_timeline.py $out | tee $out/overlap_timeline.md
rm -rf $out/plain $out/persistent $out/per_tile
