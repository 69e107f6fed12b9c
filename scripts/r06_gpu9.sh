#!/bin/bash
# round 6: the rotary embedding's backward inside the attention backward's epilogues — tests, then the step with the switch
# off / on, interleaved on one box
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_fused_gpu.py tests/test_full_size_parity_gpu.py tests/test_reference_fixtures_gpu.py -q -x -m gpu -k "attention or attn or rope or rotary or parity or fixture or golden" > gpurun_out/r06_rope_grad_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_rope_grad_tests.log
tail -4 gpurun_out/r06_rope_grad_tests.log
: > gpurun_out/r06_rope_grad_ab.log
for rep in 1 2 3; do
  for sw in 0 1; do
    TN_ROPE_GRAD_IN_ATTENTION=$sw python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rotary gradient in the attention backward $sw:', d['ms_per_step'], 'ms  loss', d['loss_per_sample_last'])" | tee -a gpurun_out/r06_rope_grad_ab.log
  done
done
