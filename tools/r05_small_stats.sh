#!/bin/bash
# Kernel stats of the small-batch path at batch B (GPU box, repo root):  bash tools/r05_small_stats.sh <B> [precision]
B=${1:-1}; P=${2:-bf16}; REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/sb_run.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from bench import build_model
from beso_amd import _lib, synthetic as S
from beso_amd.runtime import set_plan
cfg = S.SHAPES["kitchen"]
m = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), "$P", "cuda:0")
s, g, a = (torch.from_numpy(v).to("cuda:0") for v in S.make_inputs(cfg, $B, seed=1))
sg = torch.full(($B,), 0.3, device="cuda:0")
set_plan(forward=_lib.PLAN_SMALL)
with torch.no_grad():
    for _ in range(200): m(s, a, g, sg)
torch.cuda.synchronize()
PY
cd /tmp; rm -rf $O/prof_sb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sb -o tr -- python /tmp/sb_run.py > /dev/null 2>&1
cd $REPO
f=$(find $O/prof_sb -name "*kernel_stats.csv" | head -1)
python tools/kernel_stats.py $f 200 14 | cut -c1-160
t=$(find $O/prof_sb -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
# one forward in steady state: 32 kernels from an embed kernel
idx = [i for i, r in enumerate(rows) if "embed" in r[2]]
a = idx[-3]; b = idx[-2]
t0 = rows[a][0]
for s, e, n in rows[a:b]:
    print(f"{(s - t0) / 1000:8.1f} us  dur {(e - s) / 1000:6.1f}  {n}")
print(f"forward: {(rows[b][0] - t0) / 1000:.1f} us start to start")
PY
rm -rf $O/prof_sb
