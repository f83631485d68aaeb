#!/bin/bash
# Kernel stats and the kernel timeline of one BesoAgent.predict() call at one environment (GPU box, repo root)
REPO=$(pwd); O=$REPO/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/pred_run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tools")
from bench import build_model
from beso_amd import synthetic as O
from _agent import build_agent
from beso_amd.networks.scaler.scaler_class import Scaler
dev = "cuda:0"
cfg = O.SHAPES["kitchen"]
w = O.make_weights(cfg, seed=0, std=0.02)
agent = build_agent(cfg, lambda: build_model(cfg, w, "bf16", dev), device=dev)
rng = np.random.default_rng(0)
agent.get_scaler(Scaler(rng.standard_normal((256, cfg.obs_dim)).astype(np.float32), rng.standard_normal((256, cfg.act_dim)).astype(np.float32), True, dev))
agent.set_bounds(agent.scaler)
agent.reset()
goal = torch.randn(cfg.goal_seq_len, cfg.obs_dim)
for _ in range(120): agent.predict({"observation": torch.randn(1, cfg.obs_dim), "goal_observation": goal})
torch.cuda.synchronize()
PY
cd /tmp; rm -rf $O/prof_pred
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pred -o tr -- python /tmp/pred_run.py > /dev/null 2>&1
cd $REPO
python tools/kernel_stats.py $(find $O/prof_pred -name "*kernel_stats.csv" | head -1) 120 24 | cut -c1-170
python - "$(find $O/prof_pred -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
idx = [i for i, r in enumerate(rows) if "embed_kernel" in r[2]]
# one predict call in steady state = three embeds: from the launch after the head kernel before the 3rd-last group
a = idx[-7]; b = idx[-4]
t0 = rows[a][0]
# back up to the kernels between the previous call's last head and this call's first embed
k = a
while k > 0 and "head_kernel" not in rows[k - 1][2]: k -= 1
for s, e, n in rows[k:b]:
    if "sb_" in n: continue
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:6.1f}  {n}")
PY
