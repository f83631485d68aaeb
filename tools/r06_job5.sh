#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 200 python tools/fuzz_small.py 40 22 2>&1 | tail -2
timeout 600 python tools/latency_small.py kitchen 2>&1 | grep -v amdgpu.ids | tee $O/r06_latency_small_b.txt | grep bf16
timeout 300 bash tools/r05_small_stats.sh 1 2>&1 | head -8 | cut -c1-150
timeout 900 python tools/r06_explain_lnf.py 11 2>&1 | grep -v amdgpu.ids | tee $O/r06_explain_lnf.txt
