#!/usr/bin/env python3
"""Development aid: wall time of BesoAgent.predict (one environment step: window update, 3-step DDIM over
the action window, clip, inverse scale) for B parallel environments, next to the bare sampler call.
Run on the GPU box:  python tools/latency_predict.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import synthetic as O  # noqa: E402
from _agent import build_agent  # noqa: E402
from beso_amd.networks.scaler.scaler_class import Scaler  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks  # noqa: E402


def main():
    dev = "cuda:0"
    cfg = O.SHAPES["kitchen"]
    w = O.make_weights(cfg, seed=0, std=0.02)
    agent = build_agent(cfg, lambda: build_model(cfg, w, "bf16", dev), device=dev)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((256, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((256, cfg.act_dim)).astype(np.float32), True, dev))
    agent.set_bounds(agent.scaler)
    for B in (1, 16, 256):
        agent.reset()
        goal = torch.randn(cfg.goal_seq_len, cfg.obs_dim)
        obs = [torch.randn(B, cfg.obs_dim) for _ in range(40)]
        for o in obs[:10]:
            agent.predict({"observation": o, "goal_observation": goal})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in obs[10:]:
            out = agent.predict({"observation": o, "goal_observation": goal})
        torch.cuda.synchronize()
        per_call = (time.perf_counter() - t0) / 30
        # the bare 3-step sampler on resident tensors
        s, g, a = (torch.from_numpy(v).to(dev) for v in O.make_inputs(cfg, B, seed=1))
        sig = ks.get_sigmas_exponential(3, 0.005, 1.0)
        with torch.no_grad():
            for _ in range(5):
                ks.sample_ddim(agent.model, s, a, g, sig, disable=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                ks.sample_ddim(agent.model, s, a, g, sig, disable=True)
            torch.cuda.synchronize()
            bare = (time.perf_counter() - t0) / 30
        print(f"B={B:4d}: predict {per_call * 1e3:7.3f} ms/call   3-step DDIM alone {bare * 1e3:7.3f} ms   "
              f"(out {tuple(out.shape)})", flush=True)


if __name__ == "__main__":
    main()
