#!/bin/bash
# same-box A/B of the small-batch path: product (12-wave attention launch + out-projection epilogue) against round 5's library
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
V=$REPO/beso_amd/lib/variants
for rep in 1 2; do
for lib in "" $V/libbeso_hip_r5.so; do
  name=$(basename "${lib:-product}" .so)
  for B in 1 4 8 16; do
    BESO_HIP_LIB=$lib timeout 200 python tools/r05_graph_small.py $B 2>&1 | grep -v amdgpu.ids | sed "s/^/$name /"
  done
  BESO_HIP_LIB=$lib timeout 300 python tools/latency_predict.py 2>&1 | grep "B=" | sed "s/^/$name /"
done; done 2>&1 | tee $O/r06_small_ab.txt
