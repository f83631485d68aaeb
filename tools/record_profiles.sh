#!/bin/bash
# Record run for profiles/: bench line + rocprofv3 kernel stats + separate PMC passes.
# Run on the GPU box from the repo root:   bash tools/record_profiles.sh r01
# Outputs land in gpurun_out/ (scratch); tools/pmc_summary.py turns them into profiles/.
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-other-configs"       # (bench.py runs its own PMC passes otherwise)
BENCHP="$BENCH --no-parity-line"                                                         # counter passes: the bf16 instance only
cd /tmp
rm -rf $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write $OUT/prof_${TAG}_sq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -o $TAG -- $BENCH > $OUT/rocprof_${TAG}_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -o $TAG -- $BENCHP > $OUT/rocprof_${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -o $TAG -- $BENCHP > $OUT/rocprof_${TAG}_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_${TAG}_sq -o $TAG -- $BENCHP > $OUT/rocprof_${TAG}_sq.log 2>&1
cd $REPO
# traffic.json first (bench.py quotes it), then the bench line of the same box
python tools/pmc_summary.py $TAG --out gpurun_out/profiles_$TAG
cp gpurun_out/profiles_$TAG/traffic.json profiles/traffic.json
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
python tools/pmc_summary.py $TAG --out gpurun_out/profiles_$TAG
