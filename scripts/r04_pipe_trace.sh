#!/bin/bash
# timeline of the pipelined optimizer update: which kernels run while the adamw launches are on the device, and how long
R=$(pwd); out=$R/gpurun_out/r04q; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
for pipe in 1 0; do
TN_PIPELINE_OPTIMIZER=$pipe rocprofv3 --kernel-trace -d $out/prof$pipe --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-rooflines > $out/prof$pipe.log 2>&1
f=$(ls $out/prof$pipe/*/*kernel_trace.csv | head -1)
python3 $R/scripts/r04_pipe_trace.py $f | tee $out/timeline_pipe$pipe.log
rm -rf $out/prof$pipe
done
