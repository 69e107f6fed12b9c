cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "\(TCC\|TCP\|TA\|TD\)_[A-Za-z0-9_]*" | sort -u > $R/gpurun_out/counters_avail.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum" "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/gemm_mem_pmc/p$i --output-format csv -- python $R/scripts/gemm_prof.py 3 > $R/gpurun_out/gemm_mem_pmc_p$i.log 2>&1
done
cd $R; python scripts/pmc_sum.py gpurun_out/gemm_mem_pmc --match emm; python scripts/pmc_sum.py gpurun_out/gemm_mem_pmc --match Cijk
tail -3 gpurun_out/gemm_mem_pmc_p*.log | head -40
