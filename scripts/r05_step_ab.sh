#!/bin/bash
# Round 5: whole-step A/B on ONE box (interleaved), then the kernel statistics of the default configuration.
#   usage: bash scripts/r05_step_ab.sh <tag> "<name>=<ENV ...>" ...     (first configuration = default; "prof" as last arg adds rocprof)
R=$(pwd); tag=$1; shift; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
run() {
  env $2 timeout 900 python bench.py --steps 8 --warmup 3 --no-kernel-rooflines --no-cpu-baseline > $out/b.json 2> $out/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/b.json").read().strip().splitlines()[-1])
    print("$1", d["ms_per_step"], "ms", d["value"], "tok/s loss", d.get("loss_per_sample_last"), "mem", d.get("peak_mem_GB_rank0"), "frac", d["roofline"]["frac"])
except Exception as e:
    print("$1 failed", e); print(open("$out/b.err").read()[-2500:])
PY
}
prof=0
cfgs=()
for a in "$@"; do if [ "$a" = prof ]; then prof=1; else cfgs+=("$a"); fi; done
for rep in 1 2; do
  for c in "${cfgs[@]}"; do run "${c%%=*}" "${c#*=}"; done
done 2>&1 | tee $out/summary.log
if [ $prof = 1 ]; then
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/prof.log 2>&1
  cd $R
  f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
  python scripts/summarize_rocprof.py $f $out/qwen2audio7b_kernel_stats.md > /dev/null && head -60 $out/qwen2audio7b_kernel_stats.md | cut -c1-200
  rm -rf $out/prof
fi
