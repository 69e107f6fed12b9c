#!/bin/bash
# Round 6: PMC counters of the final attention kernels on the headline shapes (scripts/r06_attn_final_wl.py).
#   usage (GPU box, repo root): bash scripts/r06_attn_final_pmc.sh  -> gpurun_out/r06_attn_final_pmc.md
R=$(pwd); cd /tmp; export TMPDIR=/tmp; out=$R/gpurun_out/r06_final_pmc; rm -rf $out
run() { rocprofv3 --kernel-trace --pmc "$@" -d $out/$tag --output-format csv -- python $R/scripts/r06_attn_final_wl.py > /dev/null 2>&1; }
tag=p1; run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
tag=p2; run SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE
tag=p3; run TCC_HIT_sum TCC_MISS_sum
tag=p4; run FETCH_SIZE
tag=p5; run WRITE_SIZE
cd $R; python scripts/pmc_sum.py $out --match attn_ > gpurun_out/r06_attn_final_pmc.md; cat gpurun_out/r06_attn_final_pmc.md
rm -rf $out
