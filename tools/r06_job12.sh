#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
V=$REPO/beso_amd/lib/variants
timeout 900 python -m pytest tests -m gpu -x -q -k "train or loss or grad or scaler or agent" 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 200 python tools/fuzz_train_bf16.py 60 12 2>&1 | tail -1 | cut -c1-300
for rep in 1 2; do
for lib in "" $V/libbeso_hip_r6a.so; do
  name=$(basename "${lib:-product}" .so)
  for cfg in "1024 kitchen" "8192 kitchen" "1024 block_push"; do
    r=$(BESO_HIP_LIB=$lib timeout 300 python tools/bench_train.py $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % (d['seconds_per_step']*1e3))")
    echo "$name $cfg: $r"
  done
done; done 2>&1 | tee $O/r06_train_ab_tail.txt
