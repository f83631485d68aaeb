#!/bin/bash
# Same-box A/B of ONE kernel of the training step over the built variants (tools/variants.py build ...):
#   bash tools/r05_kernel_ab.sh <kernel substring> [batch] [shape]        (GPU box, repo root)
SUB=$1; B=${2:-1024}; SHAPE=${3:-kitchen}; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" $(ls $REPO/beso_amd/lib/variants/*.so 2>/dev/null); do
  name=$(basename "${lib:-product}" .so)
  cd /tmp; rm -rf $O/ab_prof
  BESO_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab_prof -o tr -- python $REPO/tools/bench_train.py $B $SHAPE > $O/ab_line.txt 2>&1
  cd $REPO
  f=$(find $O/ab_prof -name "*kernel_stats.csv" | head -1)
  step=$(grep -o '"seconds_per_step": [0-9.e-]*' $O/ab_line.txt | tail -1)
  python - "$f" "$SUB" "$name" "$step" <<'PY'
import csv, sys
f, sub, name, step = sys.argv[1:5]
out = []
for r in csv.DictReader(open(f)):
    if sub in r["Name"]:
        out.append(f'{r["Name"].split("(")[0][-40:]} x{r["Calls"]} avg {float(r["AverageNs"]) / 1000:.1f} us')
print(f"{name:20s} {'; '.join(out)}  {step}")
PY
done; done
rm -rf $O/ab_prof $O/ab_line.txt
