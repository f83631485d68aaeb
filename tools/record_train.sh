#!/bin/bash
# Training-step record run for profiles/: bench lines, rocprofv3 kernel stats and three PMC passes (each its own run).
# On the GPU box from the repo root:   bash tools/record_train.sh r03
TAG=${1:-r03}; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python tools/bench_train.py 1024 > $O/train_hip.txt 2>&1
python tools/bench_train.py 8192 > $O/train_hip8k.txt 2>&1
python tools/bench_train.py 1024 kitchen --autograd > $O/train_eager.txt 2>&1
cd /tmp
rm -rf $O/prof_train_stats $O/pmc_train_a $O/pmc_train_b $O/pmc_train_c
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_stats -o tr -- python $REPO/tools/bench_train.py 1024 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_train_a -o tr -- python $REPO/tools/bench_train.py 1024 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_train_b -o tr -- python $REPO/tools/bench_train.py 1024 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_train_c -o tr -- python $REPO/tools/bench_train.py 1024 > /dev/null 2>&1
cd $REPO
python tools/pmc_train.py $O/pmc_train_a $O/pmc_train_b $O/pmc_train_c > $O/${TAG}_train_step_pmc.json
