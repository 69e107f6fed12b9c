#!/bin/bash
# Round 5 (round 4: scripts/r04_config_d_memory.sh): config D (Qwen2-Audio-7B long audio, T = 65536, cp = 4) on ONE emulated rank whose optimizer
# state and gradient buckets are sharded like rank 0 of the dp x cp = 8 group the configuration names (flat engine,
# collectives replaced by local copies): peak memory and step time for activation checkpointing none / op-level (row kernels recomputed) / every 2nd block / full.
out=gpurun_out/${1:-r05m}; mkdir -p $out; export TMPDIR=/tmp
for ac in none op selective full; do
  timeout 900 python bench.py --workload qwen2_audio_7b_long --cp 4 --emulate-rank 0 --emulate-shards 8 --ac $ac --steps 3 --warmup 2 \
      --no-cpu-baseline --no-kernel-rooflines > $out/d_ac_$ac.json 2> $out/d_ac_$ac.err
  python3 - $out/d_ac_$ac.json $ac <<'PY' | tee -a $out/summary.log
import json, sys
try:
    l = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print(f"config D rank 0 (state sharded 1/8), AC {sys.argv[2]}: {l['ms_per_step']} ms/step, peak {l['peak_mem_GB_rank0']} GB, executed-FLOP frac {l.get('step_mfu_executed_flops')}, loss {l['loss_per_sample_last']}")
except Exception as e:
    print("AC", sys.argv[2], "failed:", e)
PY
done
