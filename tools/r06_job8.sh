#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bash tools/r06_job7.sh 2>&1 | grep -v "B= 256\|B=8 \|B=4 "
timeout 200 python tools/fuzz_small.py 30 24 2>&1 | tail -1
bash tools/r06_x3_stats.sh 2>&1 | tee $O/r06_x3_stats.txt
