#!/bin/bash
# Round 5: one bench line per workload on one MI355X (emulated ranks for configs D / E) -> gpurun_out/<tag>/
out=gpurun_out/${1:-r05w}; mkdir -p $out; export TMPDIR=/tmp
run() {   # name, args...
  local name=$1; shift
  timeout 900 python bench.py "$@" --no-cpu-baseline --no-kernel-rooflines > $out/$name.json 2> $out/$name.err
  python3 - $out/$name.json $name <<'PY' | tee -a $out/summary.log
import json, sys
try:
    l = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    r = l.get("roofline", {})
    print(f"{sys.argv[2]}: {l['ms_per_step']} ms/step, {l['value']} tokens/s, formula MFU {l.get('step_mfu')}, executed-FLOP frac {r.get('frac')}, peak {l.get('peak_mem_GB_rank0')} GB, loss {l.get('loss_per_sample_last')}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run headline_a --steps 8 --warmup 3
run headline_b --steps 8 --warmup 3
run llama_asr_1b --workload llama_asr_1b --steps 20 --warmup 5
run config_D_cp4_rank0 --workload qwen2_audio_7b_long --cp 4 --emulate-rank 0 --emulate-shards 8 --steps 3 --warmup 2
run config_D_cp4_rank3 --workload qwen2_audio_7b_long --cp 4 --emulate-rank 3 --emulate-shards 8 --steps 3 --warmup 2
run config_E_tp2_rank0 --workload kimi_audio_7b --tp 2 --emulate-rank 0 --steps 6 --warmup 2
run config_E_tp2_rank0_loss_parallel --workload kimi_audio_7b --tp 2 --emulate-rank 0 --loss-parallel --steps 6 --warmup 2
run config_E_speech_tp2_rank0 --workload kimi_audio_7b_speech --tp 2 --emulate-rank 0 --steps 5 --warmup 2
