#!/bin/bash
# Counters of ONE kernel of the training step (GPU box, repo root):  bash tools/r05_pmc_kernel.sh <kernel substring> <tag> [batch] [shape]
SUB=$1; TAG=$2; B=${3:-1024}; SHAPE=${4:-kitchen}; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rm -rf $O/pk_a $O/pk_b $O/pk_c $O/pk_d $O/pk_e
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pk_a -o p -- python $REPO/tools/bench_train.py $B $SHAPE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pk_b -o p -- python $REPO/tools/bench_train.py $B $SHAPE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pk_c -o p -- python $REPO/tools/bench_train.py $B $SHAPE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pk_d -o p -- python $REPO/tools/bench_train.py $B $SHAPE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pk_e -o p -- python $REPO/tools/bench_train.py $B $SHAPE > /dev/null 2>&1
cd $REPO
python tools/pmc_kernel.py "$SUB" $O/pk_a $O/pk_b $O/pk_c $O/pk_d $O/pk_e > $O/${TAG}_pmc.json
rm -rf $O/pk_a $O/pk_b $O/pk_c $O/pk_d $O/pk_e
cat $O/${TAG}_pmc.json
