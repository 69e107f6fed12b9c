#!/bin/bash
# pipelined optimizer update: GPU test + headline sweep over the side launch's workgroup bound on one box
out=gpurun_out/r04p; mkdir -p $out
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -k "optimizer_updates_under or fused_adamw" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
run() {
  TN_PIPELINE_OPTIMIZER=$1 TN_ADAMW_SIDE_WORKGROUPS=$2 timeout 900 python bench.py --steps 8 --warmup 3 --no-kernel-rooflines --no-cpu-baseline > $out/b.json 2> $out/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/b.json").read().strip().splitlines()[-1])
    print("pipe=$1 wgs=$2", d["ms_per_step"], "ms", d["value"], "tok/s peak", d.get("peak_mem_GB_rank0"), "loss", d.get("loss_per_sample_last"))
except Exception as e:
    print("pipe=$1 wgs=$2 failed", e); print(open("$out/b.err").read()[-2000:])
PY
}
{
run 0 0
for w in ${SWEEP:-16 32 64 128 256 512}; do run 1 $w; done
run 0 0
} 2>&1 | tee $out/summary.log
