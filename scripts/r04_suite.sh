#!/bin/bash
# full GPU suite + smoke + the config-E speech workload on an emulated tp rank
tag=${1:-r04s}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py --workload kimi_audio_7b_speech --tp 2 --emulate-rank 0 --steps 5 --warmup 2 --no-kernel-rooflines --no-cpu-baseline > $out/kimi_speech_tp2_rank0.json 2> $out/kimi_speech.err
echo "kimi rc=$?"; cut -c1-600 $out/kimi_speech_tp2_rank0.json; tail -5 $out/kimi_speech.err
timeout 900 python bench.py --workload kimi_audio_7b --tp 2 --emulate-rank 0 --steps 5 --warmup 2 --no-kernel-rooflines --no-cpu-baseline > $out/kimi_tp2_rank0.json 2>> $out/kimi_speech.err
cut -c1-400 $out/kimi_tp2_rank0.json
