#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/sbs_run.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from bench import build_model
from beso_amd import synthetic as S
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
cfg = S.SHAPES["kitchen"]
m = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), "bf16", "cuda:0")
s, g, a = (torch.from_numpy(v).to("cuda:0") for v in S.make_inputs(cfg, 1, seed=1))
sig = ks.get_sigmas_exponential(3, 0.005, 1.0)
with torch.no_grad():
    for _ in range(100): ks.sample_ddim(m, s, a, g, sig, disable=True)
torch.cuda.synchronize()
PY
cd /tmp; rm -rf $O/prof_sbs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sbs -o tr -- python /tmp/sbs_run.py > /dev/null 2>&1
cd $REPO
f=$(find $O/prof_sbs -name "*kernel_stats.csv" | head -1)
python tools/kernel_stats.py $f 100 16 | cut -c1-160
