#!/bin/bash
TAG="default" python scripts/r04_determinism2.py 250 2>&1 | grep -v amdgpu.ids
TAG="default again" python scripts/r04_determinism2.py 250 2>&1 | grep -v amdgpu.ids
for i in 1 2 3 4 5 6; do python -m pytest tests/test_parallel_gpu.py -x -q -k "side_stream_train_identically" 2>&1 | grep -E "passed|failed|AssertionError: " | cut -c1-300; done
