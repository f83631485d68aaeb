#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16x3 or long" -s 2>&1 | grep -E "parity\]|passed|failed|Error" | head -30
bash tools/r06_x3_stats.sh 2>&1 | head -6 | tee $O/r06_x3_stats_after.txt
timeout 600 python tools/bench_configs.py --out $O/r06_configs_x3.json 2>&1 | grep -v amdgpu | tail -15
