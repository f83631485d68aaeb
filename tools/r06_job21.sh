#!/bin/bash
# Kept rows of the training forward as 16-byte pieces wherever two tiles pair up (LayerNorm outputs, q|k|v, y, bf16 residuals):
# parity, kernel stats and the step's same-box A/B against the commit before ('head')
O=gpurun_out; mkdir -p $O
unset BESO_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train" 2>&1 | tail -2
for v in base head; do
  if [ $v = base ]; then unset BESO_HIP_LIB; else export BESO_HIP_LIB=$(pwd)/beso_amd/lib/variants/libbeso_hip_$v.so; fi
  for cfg in "1024 kitchen" "8192 kitchen" "1024 block_push"; do
    set -- $cfg
    timeout 200 bash tools/r05_train_stats.sh k16_${v}_$1_$2 $1 $2 > /dev/null 2>&1
    echo "== $v $cfg"; grep -E "train_fwd_kernel" $O/k16_${v}_$1_$2_kernel_stats.txt | head -1
  done
done
unset BESO_HIP_LIB
bash tools/r05_train_ab.sh > $O/k16_ab.txt 2>&1; cat $O/k16_ab.txt
