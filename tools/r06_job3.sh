#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for B in 64 128 256; do echo "=== B=$B"; timeout 300 bash tools/r05_small_stats.sh $B 2>&1 | head -8; done 2>&1 | tee $O/r06_small_mid_before.txt
