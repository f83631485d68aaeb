#!/bin/bash
# Store-cost experiments of the training forward (GPU box, repo root): variants built by tools/variants.py
O=gpurun_out; mkdir -p $O
for v in "$@"; do
  if [ $v = base ]; then unset BESO_HIP_LIB; else export BESO_HIP_LIB=$(pwd)/beso_amd/lib/variants/libbeso_hip_$v.so; fi
  B=${BATCH:-1024}
  timeout 120 bash tools/r05_train_stats.sh fold_$v $B kitchen > $O/fold_$v.txt 2>&1
  echo "== $v"; grep -E "train_fwd_kernel" $O/fold_${v}_kernel_stats.txt | head -2
done
