#!/usr/bin/env python3
"""Small batches: one GCDenoiser.forward and short sampler calls through the chip-wide small-batch path (the library's choice
up to 448 (bf16) / 4096 (fp32) token rows; small.hip) against the one-launch kernel's latency instance (BESO_PLAN_FUSED), bf16 and fp32, with the
deviation of the two from each other:   python tools/latency_small.py [kitchen|block_push]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import _lib, synthetic as S  # noqa: E402
from beso_amd.runtime import set_plan  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks  # noqa: E402


def timed(fn, n=200, warm=20):
    with torch.no_grad():
        for _ in range(warm):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
    dev = "cuda:0"
    cfg = S.SHAPES[name]
    w = S.make_weights(cfg, seed=0, std=0.02)
    for prec in ("bf16", "fp32"):
        model = build_model(cfg, w, prec, dev)
        for B in (1, 2, 4, 16, 32, 64, 93, 128, 256):
            s, g, a = (torch.from_numpy(v).to(dev) for v in S.make_inputs(cfg, B, seed=1))
            sg = torch.full((B,), 0.3, device=dev)
            row = {}
            for label, hint in (("small", _lib.PLAN_SMALL), ("default", 0), ("one-launch" if prec == "bf16" else "per-op",
                                                                             _lib.PLAN_FUSED if prec == "bf16" else _lib.PLAN_PER_OP)):
                set_plan(forward=hint)
                row[label] = timed(lambda: model(s, a, g, sg))
            set_plan(forward=0)
            other = "one-launch" if prec == "bf16" else "per-op"
            dev_rel = ((row["small"][1] - row[other][1]).abs().max() / row[other][1].abs().max()).item()
            print(f"{name} {prec} B={B:4d} forward: small path {row['small'][0]:7.1f} us   library's choice {row['default'][0]:7.1f} us   "
                  f"{other} {row[other][0]:7.1f} us   max rel deviation small vs {other} {dev_rel:.1e}", flush=True)
        # BASELINE configs[0]: B = 64, 10 DDIM steps; the rollout call: B = 1, 3 steps of euler_ancestral / DDIM
        for B, steps in ((64, 10), (1, 3), (16, 3)):
            s, g, a = (torch.from_numpy(v).to(dev) for v in S.make_inputs(cfg, B, seed=1))
            sig = ks.get_sigmas_exponential(steps, 0.005, 1.0)
            for label, hint in (("library's choice", 0), ("one-launch loop" if prec == "bf16" else "per-op", _lib.PLAN_FUSED if prec == "bf16" else _lib.PLAN_PER_OP)):
                set_plan(forward=hint)
                td, _ = timed(lambda: ks.sample_ddim(model, s, a, g, sig, disable=True), n=50, warm=5)
                te, _ = timed(lambda: ks.sample_euler_ancestral(model, s, a, g, sig, disable=True), n=50, warm=5)
                print(f"{name} {prec} B={B:3d} {steps:2d} steps, {label}: sample_ddim {td / 1e3:6.3f} ms   sample_euler_ancestral {te / 1e3:6.3f} ms", flush=True)
            set_plan(forward=0)


if __name__ == "__main__":
    main()
