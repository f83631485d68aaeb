#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "small" 2>&1 | tail -5
timeout 200 python tools/fuzz_small.py 60 21 2>&1 | tail -3
timeout 600 python tools/latency_small.py kitchen 2>&1 | grep -v amdgpu.ids | tee $O/r06_latency_small_a.txt
for B in 64 128; do echo "=== B=$B"; timeout 300 bash tools/r05_small_stats.sh $B 2>&1 | head -8; done 2>&1 | tee $O/r06_small_mid_after.txt
