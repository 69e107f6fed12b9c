#!/bin/bash
mkdir -p gpurun_out
HPW=1 timeout 300 python scripts/r06_attn_trace.py > gpurun_out/r06_attn_trace_hpw1.log 2>&1
cat gpurun_out/r06_attn_trace_hpw1.log
