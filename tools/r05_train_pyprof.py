#!/usr/bin/env python3
"""cProfile of BesoAgent.train_step's host side:  python tools/r05_train_pyprof.py [batch] [shape]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from bench import build_model
from beso_amd import synthetic as O
from _agent import build_agent
from beso_amd.networks.scaler.scaler_class import Scaler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
name = sys.argv[2] if len(sys.argv) > 2 else "kitchen"
dev = "cuda:0"; cfg = O.SHAPES[name]; w = O.make_weights(cfg, seed=0, std=0.02)
attn_p, resid_p = {"kitchen": (0.3, 0.0), "block_push": (0.05, 0.05)}[name]
def model():
    m = build_model(cfg, w, "bf16", dev); m.inner_model._pdrops = (0.0, attn_p, resid_p); return m
agent = build_agent(cfg, model, device=dev, lr=1e-4)
rng = np.random.default_rng(0)
agent.get_scaler(Scaler(rng.standard_normal((64, cfg.obs_dim)).astype(np.float32), rng.standard_normal((64, cfg.act_dim)).astype(np.float32), True, dev))
agent.set_bounds(agent.scaler)
batch = {"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim, device=dev), "action": torch.tanh(torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=dev)),
         "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim, device=dev)}
for _ in range(30): agent.train_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): agent.train_step(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(38)
