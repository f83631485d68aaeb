#!/usr/bin/env python3
"""Development aid: latency of one GCDenoiser.forward and of a 3-step DDIM loop vs batch size, with the
fused kernel on / off (BESO_PLAN_PER_OP hint).  Run on the GPU box:  python tools/latency.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(min_batch):
    sys.path.insert(0, ROOT)
    import torch
    from beso_amd import _lib
    from beso_amd.runtime import set_plan
    set_plan(forward=_lib.PLAN_PER_OP if min_batch else 0)
    from bench import build_model
    from beso_amd import synthetic as O
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    dev = "cuda:0"
    cfg = O.SHAPES["kitchen"]
    model = build_model(cfg, O.make_weights(cfg, seed=0, std=0.02), "bf16", dev)
    sig3 = ks.get_sigmas_exponential(3, 0.005, 1.0)
    for B in (1, 8, 32, 64, 128, 256, 512, 1024, 4096):
        s, g, a = (torch.from_numpy(v).to(dev) for v in O.make_inputs(cfg, B, seed=1))
        sg = torch.full((B,), 0.3, device=dev)
        with torch.no_grad():
            for _ in range(5):
                model(s, a, g, sg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 50
            for _ in range(n):
                model(s, a, g, sg)
            torch.cuda.synchronize()
            fwd = (time.perf_counter() - t0) / n
            for _ in range(3):
                ks.sample_ddim(model, s, a, g, sig3, disable=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                ks.sample_ddim(model, s, a, g, sig3, disable=True)
            torch.cuda.synchronize()
            ddim = (time.perf_counter() - t0) / n
        print(f"min_batch={min_batch:6d} B={B:5d}  forward {fwd * 1e6:8.1f} us   ddim3 {ddim * 1e6:8.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for mb in (0, 1 << 30):
            subprocess.run([sys.executable, __file__, str(mb)], check=True)
