#!/bin/bash
# Ablation timings (variant libraries abl1..abl6 from scripts/build_gemm_variant.sh ablN -DTN_GEMM_ABLATE=N) and SQ
# counters of the hand-written GEMM vs the library kernel.  Run on the GPU box from the repo root.
R=$PWD; mkdir -p gpurun_out
SH="fwd   q/o,fwd   down,wgrad q/o"
echo "== full kernel"; python scripts/gemm_sweep.py --no-check --variants 101 --shapes "$SH" --rounds 3 --iters 10 2>&1 | grep -v "^ *best"
for n in 1 2 3 4 5 6; do
  echo "== ablation $n"
  TN_AMD_LIB=$R/touchnet_amd/_lib/variants/abl$n/libtouchnet_amd.so python scripts/gemm_sweep.py --no-check --variants 101 --shapes "$SH" --rounds 3 --iters 10 2>&1 | grep -v "^ *best"
done
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/gemm_pmc_r03/p$i --output-format csv -- python $R/scripts/gemm_prof.py 3 > $R/gpurun_out/gemm_pmc_r03_p$i.log 2>&1
done
cd $R; python scripts/pmc_sum.py gpurun_out/gemm_pmc_r03 --match gemm_kernel; python scripts/pmc_sum.py gpurun_out/gemm_pmc_r03 --match Cijk
