#!/bin/bash
# Round-6 record run for profiles/ (GPU box, repo root; beso_amd/lib/variants must hold st2, st3 and r5 = round 5's library:
#   python tools/variants.py build st2=-DBESO_DEV_API=1,-DBESO_FUSED_STAMPS=2 st3=-DBESO_DEV_API=1,-DBESO_FUSED_STAMPS=3 r5=@7162316):
#   bash tools/record_r06.sh
TAG=r06; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
V=$REPO/beso_amd/lib/variants
bash tools/record_profiles.sh $TAG
python bench.py --workload train --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_train.json
# training step: bench lines, kernel stats, counters, phase stamps, same-box A/B against round 5's library
bash tools/record_train.sh $TAG
f=$(find $O/prof_train_stats -name "*kernel_stats.csv" | head -1); cp $f $O/${TAG}_train_step_kernel_stats.csv
python tools/kernel_stats.py $f 13 40 > $O/${TAG}_train_step_kernel_stats.txt
python tools/bench_train.py 1024 block_push 2>&1 | tail -1 > $O/${TAG}_train_block_push.json
bash tools/r05_train_stats.sh ${TAG}_train_bp 1024 block_push > /dev/null 2>&1
bash tools/r05_train_stats.sh ${TAG}_train_8k 8192 kitchen > /dev/null 2>&1
for B in 1024 8192; do
  BESO_HIP_LIB=$V/libbeso_hip_st2.so timeout 300 python tools/train_stamps.py $B 2>&1 | grep -v amdgpu.ids
done > $O/${TAG}_mlp_bwd_stamps.txt
for B in 1024 8192; do
  BESO_HIP_LIB=$V/libbeso_hip_st3.so timeout 300 python tools/train_stamps.py $B 2>&1 | grep -v amdgpu.ids
done > $O/${TAG}_dgrad_stamps.txt
for rep in 1 2; do
for lib in "" $V/libbeso_hip_r5.so; do
  name=$(basename "${lib:-product}" .so)
  for cfg in "1024 kitchen" "8192 kitchen" "1024 block_push"; do
    r=$(BESO_HIP_LIB=$lib timeout 300 python tools/bench_train.py $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % (d['seconds_per_step']*1e3))")
    echo "$name $cfg: $r"
  done
done; done > $O/${TAG}_train_ab.txt 2>&1
# small batches: latency tables, same-box A/B, kernel stats
( python tools/latency_small.py kitchen; python tools/latency_small.py block_push; python tools/latency_predict.py; python tools/r05_graph_small.py 1; python tools/r05_graph_small.py 16; python tools/r05_fp32_cross.py | grep "^fp32" ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_latency.txt
bash tools/r06_job7.sh > /dev/null 2>&1; cp $O/r06_small_ab.txt $O/${TAG}_small_ab.txt
( bash tools/r05_small_stats.sh 1 bf16; bash tools/r05_small_stats.sh 16 bf16; bash tools/r05_small_stats.sh 1 fp32; echo '== a 3-step DDIM call at one sample'; bash tools/r05_sampler_stats.sh ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_small_batch_kernels.txt
# the 1e-4 mode
bash tools/r06_x3_stats.sh > $O/${TAG}_x3_stats.txt 2>&1
python tools/bench_configs.py --out $O/${TAG}_configs.json > $O/${TAG}_configs.log 2>&1
python tests/determinism.py --reps 8 2>&1 | grep -v amdgpu.ids | tail -12 > $O/${TAG}_determinism.txt
( python tools/fuzz_small.py 60 7; python tools/fuzz_train_bf16.py 120 11; PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tests/fuzz_shapes.py 60 13; python tools/fuzz_train.py 30; python tests/fuzz_forward.py 300 11 ) 2>&1 | grep -v amdgpu.ids | grep "fuzz\|worst" > $O/${TAG}_fuzz.txt
python tools/r06_explain_lnf.py 11 2>&1 | grep -v amdgpu.ids > $O/${TAG}_explain_lnf.txt
ls -la $O/profiles_$TAG $O/${TAG}_* | head -80
