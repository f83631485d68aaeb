#!/bin/bash
# Round-6 GPU job 1: phase stamps of the two data-gradient kernels, the training step with the new ln_reduce, training parity
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
V=$REPO/beso_amd/lib/variants
for B in 1024 8192; do
  BESO_HIP_LIB=$V/libbeso_hip_st2.so timeout 300 python tools/train_stamps.py $B 2>&1 | grep -v amdgpu.ids > $O/r06_mlp_bwd_stamps_$B.txt
  BESO_HIP_LIB=$V/libbeso_hip_st3.so timeout 300 python tools/train_stamps.py $B 2>&1 | grep -v amdgpu.ids > $O/r06_dgrad_stamps_$B.txt
done
cat $O/r06_mlp_bwd_stamps_*.txt $O/r06_dgrad_stamps_*.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "train or loss or grad" 2>&1 | tail -5
for cfg in "1024 kitchen" "8192 kitchen"; do
  timeout 300 python tools/bench_train.py $cfg 2>/dev/null | tail -1
done
timeout 400 bash tools/r05_train_stats.sh r06a 8192 2>&1 | tail -12
