#!/bin/bash
# round 6: one bench line per workload on one MI355X (scripts/r05_workloads.sh with the round's tag) + the long-sequence
# attention table
bash scripts/r05_workloads.sh r06w
timeout 900 python scripts/attn_long_bench.py > gpurun_out/r06w/attention_long_sequences.log 2>&1
tail -20 gpurun_out/r06w/attention_long_sequences.log
