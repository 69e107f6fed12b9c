#!/bin/bash
# PMC counters of the FINAL attention kernels on the headline shapes (scripts/attn_prof.py: decoder D=128 packed ~790-token
# documents + plain causal, tower D=64 T=1500): VALU per MFMA, MFMA busy, LDS bank conflicts, wait breakdown.
#   usage (GPU box, repo root): bash scripts/attn_pmc.sh [docs|causal|tower|all]  -> gpurun_out/attn_pmc_final.md
R=$(pwd); cd /tmp; export TMPDIR=/tmp; WL=${1:-all}; rm -rf $R/gpurun_out/attn_pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/attn_pmc/p1 --output-format csv -- python $R/scripts/attn_prof.py $WL > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/attn_pmc/p2 --output-format csv -- python $R/scripts/attn_prof.py $WL > /dev/null 2>&1
cd $R; python scripts/pmc_sum.py gpurun_out/attn_pmc --match attn > gpurun_out/attn_pmc_final.md; cat gpurun_out/attn_pmc_final.md
