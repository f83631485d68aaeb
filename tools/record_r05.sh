#!/bin/bash
# Round-5 record run for profiles/ (GPU box, repo root):  bash tools/record_r05.sh
TAG=r05; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bash tools/record_profiles.sh $TAG
python bench.py --workload train --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_train.json
# training step: bench lines, kernel stats, counters
bash tools/record_train.sh $TAG
f=$(find $O/prof_train_stats -name "*kernel_stats.csv" | head -1); cp $f $O/${TAG}_train_step_kernel_stats.csv
python tools/kernel_stats.py $f 13 40 > $O/${TAG}_train_step_kernel_stats.txt
python tools/bench_train.py 1024 block_push 2>&1 | tail -1 > $O/${TAG}_train_block_push.json
bash tools/r05_train_stats.sh ${TAG}_train_bp 1024 block_push > /dev/null 2>&1
bash tools/r05_train_stats.sh ${TAG}_train_8k 8192 kitchen > /dev/null 2>&1
bash tools/r05_pmc_kernel.sh wgrad_panel_group ${TAG}_wgrad 1024 > /dev/null 2>&1
bash tools/r05_pmc_kernel.sh wgrad_panel_group ${TAG}_wgrad_8k 8192 > /dev/null 2>&1
bash tools/r05_pmc_kernel.sh train_fwd_kernel ${TAG}_train_fwd 1024 > /dev/null 2>&1
( python tools/train_host_profile.py 1024; python tools/train_host_profile.py 1024 --pieces ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_train_host.txt
bash tools/r05_train_ab.sh 2>&1 | grep -v amdgpu.ids > $O/${TAG}_train_ab.txt
# small batches
( python tools/latency_small.py kitchen; python tools/latency_small.py block_push; python tools/latency_predict.py; python tools/r05_graph_small.py 1; python tools/r05_graph_small.py 16; python tools/r05_fp32_cross.py | grep "^fp32"; ./tools/microbench/launch_chain; ./tools/microbench/grid_barrier ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_latency.txt
( bash tools/r05_small_stats.sh 1 bf16; bash tools/r05_small_stats.sh 16 bf16; bash tools/r05_small_stats.sh 1 fp32; echo '== a 3-step DDIM call at one sample'; bash tools/r05_sampler_stats.sh ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_small_batch_kernels.txt
python tools/bench_configs.py --out $O/${TAG}_configs.json > $O/${TAG}_configs.log 2>&1
python tests/determinism.py --reps 8 2>&1 | grep -v amdgpu.ids | tail -12 > $O/${TAG}_determinism.txt
( python tools/fuzz_small.py 60 7; python tools/fuzz_train_bf16.py 120 11; PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tests/fuzz_shapes.py 60 13; python tools/fuzz_train.py 30; python tests/fuzz_forward.py 300 11 ) 2>&1 | grep -v amdgpu.ids | grep "fuzz\|worst" > $O/${TAG}_fuzz.txt
ls -la $O/profiles_$TAG $O/${TAG}_* | head -60
