#!/bin/bash
# attn_mfma_bwd_kernel with 128-byte LDS rows (six workgroups per CU instead of five): kernel stats at 1024 / 8192, parity
O=gpurun_out; mkdir -p $O
for v in base am64; do
  if [ $v = base ]; then unset BESO_HIP_LIB; else export BESO_HIP_LIB=$(pwd)/beso_amd/lib/variants/libbeso_hip_$v.so; fi
  for B in 1024 8192; do
    timeout 200 bash tools/r05_train_stats.sh am_${v}_$B $B kitchen > /dev/null 2>&1
    echo "== $v $B"; grep -E "attn_mfma_bwd" $O/am_${v}_${B}_kernel_stats.txt | head -1
  done
done
export BESO_HIP_LIB=$(pwd)/beso_amd/lib/variants/libbeso_hip_am64.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hip_training_step_matches_reference or training_backward_kernels or bf16_training_step_on_random" 2>&1 | tail -2
