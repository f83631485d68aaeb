#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
V=$REPO/beso_amd/lib/variants
timeout 900 python -m pytest tests -m gpu -x -q -k "train or loss or grad" 2>&1 | grep -E "passed|failed|Error|assert" | head
for rep in 1 2; do
for lib in "" $V/libbeso_hip_r6a.so; do
  name=$(basename "${lib:-product}" .so)
  for cfg in "8192 kitchen" "4096 kitchen" "1024 kitchen"; do
    r=$(BESO_HIP_LIB=$lib timeout 300 python tools/bench_train.py $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % (d['seconds_per_step']*1e3))")
    echo "$name $cfg: $r"
  done
done; done 2>&1 | tee $O/r06_train_ab_nt4.txt
timeout 400 bash tools/r05_train_stats.sh r06f 8192 2>&1 | tail -30 | head -8 | cut -c1-150
