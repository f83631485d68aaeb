#!/bin/bash
# attn_mfma_bwd_kernel with one 16-byte request per operand chunk: parity, kernel stats at 1024 / 8192 against the committed library
O=gpurun_out; mkdir -p $O
unset BESO_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train" 2>&1 | tail -2
for v in base head; do
  if [ $v = base ]; then unset BESO_HIP_LIB; else export BESO_HIP_LIB=$(pwd)/beso_amd/lib/variants/libbeso_hip_$v.so; fi
  for B in 1024 8192; do
    timeout 200 bash tools/r05_train_stats.sh am_${v}_$B $B kitchen > /dev/null 2>&1
    echo "== $v $B"; grep -E "attn_mfma_bwd|train_mlp_bwd|train_dgrad" $O/am_${v}_${B}_kernel_stats.txt | head -4
  done
  timeout 200 bash tools/r05_train_stats.sh am_${v}_bp 1024 block_push > /dev/null 2>&1
  echo "== $v block_push"; grep -E "attn_mfma_bwd|train_mlp_bwd|train_dgrad" $O/am_${v}_bp_kernel_stats.txt | head -4
done
