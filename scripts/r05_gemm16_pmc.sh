#!/bin/bash
# Round 5: LDS behaviour of Kernel16's operand modes (is the contraction-major path's smaller gain a bank-conflict problem?).
# Two PMC passes over scripts/r05_mode_bench.py (short), per-kernel means -> gpurun_out/r05_gemm16_pmc.md
#   usage (GPU box, repo root): bash scripts/r05_gemm16_pmc.sh
R=$(pwd); cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/g16pmc
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/g16pmc/p1 --output-format csv -- python $R/scripts/r05_mode_bench.py 1 2 > $R/gpurun_out/g16pmc_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $R/gpurun_out/g16pmc/p2 --output-format csv -- python $R/scripts/r05_mode_bench.py 1 2 > $R/gpurun_out/g16pmc_p2.log 2>&1
cd $R; python scripts/pmc_sum.py gpurun_out/g16pmc --match gemm > gpurun_out/r05_gemm16_pmc.md; cat gpurun_out/r05_gemm16_pmc.md
TN_GEMM_M16=0 python scripts/r05_mode_bench.py 1 3 > gpurun_out/r05_gemm16_pmc_ref32.log 2>&1
rm -rf gpurun_out/g16pmc/*/*/*.db 2>/dev/null; du -sh gpurun_out/g16pmc | tail -1
