#!/usr/bin/env python3
"""Which torch ops of a training step launch the small GPU kernels / copies around the HIP step (torch.profiler, 5 steps):
    python tools/train_small_ops.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import synthetic as O  # noqa: E402
from _agent import build_agent  # noqa: E402
from beso_amd.networks.scaler.scaler_class import Scaler  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = "cuda:0"
    cfg = O.SHAPES["kitchen"]
    w = O.make_weights(cfg, seed=0, std=0.02)

    def model():
        m = build_model(cfg, w, "bf16", dev)
        m.inner_model._pdrops = (0.0, 0.3, 0.0)
        return m

    agent = build_agent(cfg, model, device=dev, lr=1e-4)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((64, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((64, cfg.act_dim)).astype(np.float32), True, dev))
    agent.set_bounds(agent.scaler)
    batch = {"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim, device=dev),
             "action": torch.tanh(torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=dev)),
             "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim, device=dev)}
    for _ in range(10):
        agent.train_step(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        for _ in range(5):
            agent.train_step(batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))


if __name__ == "__main__":
    main()
