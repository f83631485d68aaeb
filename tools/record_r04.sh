#!/bin/bash
# Round-4 record run for profiles/ (GPU box, repo root):  bash tools/record_r04.sh
TAG=r04; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bash tools/record_profiles.sh $TAG
python tools/bench_configs.py --out $O/${TAG}_configs.json > $O/${TAG}_configs.log 2>&1
bash tools/record_train.sh $TAG
f=$(find $O/prof_train_stats -name "*kernel_stats.csv" | head -1); cp $f $O/${TAG}_train_step_kernel_stats.csv
python tools/kernel_stats.py $f 13 40 > $O/${TAG}_train_step_kernel_stats.txt
python tools/train_host_profile.py 1024 2>&1 | grep -v amdgpu.ids > $O/${TAG}_train_host.txt
BESO_AMD_ASYNC_LOSS=0 python tools/train_host_profile.py 1024 2>&1 | grep -v amdgpu.ids | head -1 | sed 's/^/BESO_AMD_ASYNC_LOSS=0: /' >> $O/${TAG}_train_host.txt
python tools/bench_train.py 1024 kitchen --per-op-forward 2>&1 | tail -1 > $O/${TAG}_train_per_op_forward.json
( python tools/latency_instances.py; python tools/latency_predict.py; python tools/latency_ancestral.py ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_latency.txt
python tests/determinism.py --reps 16 2>&1 | grep -v amdgpu.ids | tail -12 > $O/${TAG}_determinism.txt
# block-push (configs[3]): phase stamps of the sampler-loop instance and counters of its launches
BESO_HIP_LIB=$REPO/beso_amd/lib/variants/libbeso_hip_st.so python tools/phase_stamps.py 2048 block_push 2.0 2>&1 | grep -v amdgpu.ids > $O/${TAG}_block_push_stamps.txt
cd /tmp
rm -rf $O/pmc_bp_a $O/pmc_bp_b $O/pmc_bp_c
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_bp_a -o bp -- python $REPO/tools/bench_configs.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_bp_b -o bp -- python $REPO/tools/bench_configs.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_bp_c -o bp -- python $REPO/tools/bench_configs.py > /dev/null 2>&1
cd $REPO
python tools/pmc_kernel.py "layers_kernel<2, 8, 3, 4, 8, 6, 0, 0, 1>" $O/pmc_bp_a $O/pmc_bp_b $O/pmc_bp_c > $O/${TAG}_block_push_pmc.json
ls -la $O/profiles_$TAG $O/${TAG}_* | head -40
