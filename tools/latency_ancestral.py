#!/usr/bin/env python3
"""euler_ancestral (the sampler the reference's README recommends for kitchen rollouts) as one enqueue
(beso_sample_ancestral) against the step-by-step loop over the HIP denoiser.  python tools/latency_ancestral.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import synthetic as S  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks  # noqa: E402


def main():
    dev = "cuda:0"
    cfg = S.SHAPES["kitchen"]
    model = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), "bf16", dev)
    for B in (1, 64, 4096):
        s, g, a = (torch.from_numpy(v).to(dev) for v in S.make_inputs(cfg, B, seed=1))
        for n in (3, 10):
            sig = ks.get_sigmas_exponential(n, 0.005, 1.0)
            row = []
            for kw in ({}, {"callback": lambda info: None}):
                with torch.no_grad():
                    for _ in range(3):
                        ks.sample_euler_ancestral(model, s, a, g, sig, disable=True, **kw)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    reps = 30
                    for _ in range(reps):
                        ks.sample_euler_ancestral(model, s, a, g, sig, disable=True, **kw)
                    torch.cuda.synchronize()
                    row.append((time.perf_counter() - t0) / reps * 1e3)
            print(f"kitchen B={B:5d} euler_ancestral-{n:2d}: one enqueue {row[0]:7.3f} ms   step-by-step loop {row[1]:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
