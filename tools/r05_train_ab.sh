#!/bin/bash
# Same-box A/B of the training step: the product library against every variant in beso_amd/lib/variants (e.g. r4=@<round-4 commit>)
REPO=$(pwd); export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" $(ls $REPO/beso_amd/lib/variants/*.so 2>/dev/null); do
  name=$(basename "${lib:-product}" .so)
  for cfg in "1024 kitchen" "8192 kitchen" "1024 block_push"; do
    r=$(BESO_HIP_LIB=$lib python tools/bench_train.py $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % (d['seconds_per_step']*1e3))")
    echo "$name $cfg: $r"
  done
done; done
