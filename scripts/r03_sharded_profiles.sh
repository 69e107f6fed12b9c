#!/bin/bash
# rocprofv3 kernel statistics of the emulated config D (cp = 4, rank 0) and config E (tp = 2, sequence + loss parallel, rank 0) steps
R=$(pwd); mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03l_prof_D --output-format csv -- python $R/bench.py --workload qwen2_audio_7b_long --cp 4 --emulate-rank 0 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03l_prof_D.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03l_prof_E --output-format csv -- python $R/bench.py --workload kimi_audio_7b --tp 2 --loss-parallel --emulate-rank 0 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03l_prof_E.log 2>&1
cd $R
for w in D E; do
  f=$(ls gpurun_out/r03l_prof_$w/*/*kernel_trace.csv | head -1)
  python scripts/summarize_rocprof.py $f gpurun_out/r03l_config_${w}_emulated_rank0_kernel_stats.md > /dev/null && head -16 gpurun_out/r03l_config_${w}_emulated_rank0_kernel_stats.md
done
