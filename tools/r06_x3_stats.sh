#!/bin/bash
# kernel stats of the split-bf16 (1e-4) mode: long-horizon B = 256 (a 10-step Euler call) and kitchen B = 4096 forwards
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/x3_run.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from bench import build_model
from beso_amd import synthetic as S
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
name, B = sys.argv[1], int(sys.argv[2])
cfg = S.SHAPES[name]
m = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), "bf16x3", "cuda:0")
s, g, a = (torch.from_numpy(v).to("cuda:0") for v in S.make_inputs(cfg, B, seed=1))
sg = torch.full((B,), 0.3, device="cuda:0")
with torch.no_grad():
    for _ in range(20): m(s, a, g, sg)
torch.cuda.synchronize()
PY
for cfg in "long_horizon 256" "kitchen 4096"; do
  set -- $cfg
  cd /tmp; rm -rf $O/prof_x3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -o tr -- python /tmp/x3_run.py $1 $2 > /dev/null 2>&1
  cd $REPO
  f=$(find $O/prof_x3 -name "*kernel_stats.csv" | head -1)
  echo "=== $1 B=$2 bf16x3, per forward"; python tools/kernel_stats.py $f 20 14 | cut -c1-170
  rm -rf $O/prof_x3
done
