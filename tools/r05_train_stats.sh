#!/bin/bash
# Kernel stats of the training step (GPU box, repo root):  bash tools/r05_train_stats.sh <tag> [batch] [shape]
TAG=$1; B=${2:-1024}; SHAPE=${3:-kitchen}; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf $O/prof_${TAG}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -o tr -- python $REPO/tools/bench_train.py $B $SHAPE > $O/${TAG}_line.json 2>&1
cd $REPO
f=$(find $O/prof_${TAG} -name "*kernel_stats.csv" | head -1)
cp $f $O/${TAG}_kernel_stats.csv
python tools/kernel_stats.py $f 13 40 > $O/${TAG}_kernel_stats.txt
rm -rf $O/prof_${TAG}
tail -1 $O/${TAG}_line.json; head -24 $O/${TAG}_kernel_stats.txt
