#!/bin/bash
# Round-5 quick look at the training step (GPU box, repo root):  bash tools/r05_train_quick.sh [tag]
TAG=${1:-q}; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python tools/bench_train.py 1024 2>&1 | grep -v amdgpu.ids | tail -1 > $O/${TAG}_train_1024.json
python tools/bench_train.py 8192 2>&1 | grep -v amdgpu.ids | tail -1 > $O/${TAG}_train_8192.json
python tools/bench_train.py 1024 block_push 2>&1 | grep -v amdgpu.ids | tail -1 > $O/${TAG}_train_bp_1024.json
cd /tmp
rm -rf $O/prof_${TAG}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -o tr -- python $REPO/tools/bench_train.py 1024 > /dev/null 2>&1
cd $REPO
f=$(find $O/prof_${TAG} -name "*kernel_stats.csv" | head -1)
python tools/kernel_stats.py $f 13 40 > $O/${TAG}_train_kernel_stats.txt
rm -rf $O/prof_${TAG}
cat $O/${TAG}_train_1024.json $O/${TAG}_train_8192.json $O/${TAG}_train_bp_1024.json
head -30 $O/${TAG}_train_kernel_stats.txt
