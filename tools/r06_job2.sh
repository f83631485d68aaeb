#!/bin/bash
# Round-6 GPU job 2: the reworked data-gradient kernels -- parity, stamps, same-box A/B against round 5's library
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
V=$REPO/beso_amd/lib/variants
timeout 900 python -m pytest tests -m gpu -x -q -k "train or loss or grad" 2>&1 | tail -5
for B in 1024 8192; do
  BESO_HIP_LIB=$V/libbeso_hip_st2.so timeout 300 python tools/train_stamps.py $B 2>&1 | grep -v amdgpu.ids > $O/r06b_mlp_bwd_stamps_$B.txt
  BESO_HIP_LIB=$V/libbeso_hip_st3.so timeout 300 python tools/train_stamps.py $B 2>&1 | grep -v amdgpu.ids > $O/r06b_dgrad_stamps_$B.txt
done
cat $O/r06b_mlp_bwd_stamps_*.txt $O/r06b_dgrad_stamps_*.txt
for rep in 1 2; do
for lib in "" $V/libbeso_hip_r5.so; do
  name=$(basename "${lib:-product}" .so)
  for cfg in "1024 kitchen" "8192 kitchen" "1024 block_push"; do
    r=$(BESO_HIP_LIB=$lib timeout 300 python tools/bench_train.py $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % (d['seconds_per_step']*1e3))")
    echo "$name $cfg: $r"
  done
done; done 2>&1 | tee $O/r06b_train_ab.txt
timeout 400 bash tools/r05_train_stats.sh r06b 1024 2>&1 | tail -12
