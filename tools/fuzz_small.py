#!/usr/bin/env python3
"""Random model shapes / batches / windows / guidance through the chip-wide small-batch path (BESO_PLAN_SMALL) against the per-op
kernels of the same precision (BESO_PLAN_PER_OP): fp32 must agree to rounding order, bf16 inside the two-bf16-evaluations bound.
Exercises: head dims that take the head-split LN1 + q|k|v + attention launch (hd <= 64, hd % 4 == 0, T <= 16) and those that fall
back to the per-op attention kernel, ragged row tiles, K paddings (embed_dim not a multiple of 64), the MLP head, CFG pairs.
    python tools/fuzz_small.py [seconds] [seed]"""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beso_amd import _lib  # noqa: E402
from beso_amd.runtime import set_plan  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel  # noqa: E402


def run(budget=60.0, seed=0):
    rng = random.Random(seed)
    torch.manual_seed(rng.randrange(1 << 30))
    t0, n, worst = time.time(), 0, {"fp32": 0.0, "bf16": 0.0}
    while time.time() - t0 < budget:
        H = rng.choice([1, 2, 3, 4, 6, 8, 12])
        hd = rng.choice([4, 8, 12, 16, 20, 24, 32, 40, 60, 64, 72, 96])
        D = H * hd
        if D % 8 or D > 384:
            continue
        W, G = rng.randint(1, 9), rng.randint(0, 3)
        obs, act, L = rng.randint(1, 32), rng.randint(1, 16), rng.randint(1, 3)
        B, t = rng.choice([1, 2, 3, 5, 8, 17, 40, 64, 130, 300]), rng.randint(1, W)      # (64 ...: the wide bf16 instances, round 6)
        prec = rng.choice(["fp32", "bf16"])
        linear = rng.random() < 0.7
        inner = DiffusionGPT(state_dim=obs, device="cuda", goal_conditioned=G > 0, action_dim=act, embed_dim=D, embed_pdrob=0, attn_pdrop=0,
                             resid_pdrop=0, n_layers=L, n_heads=H, goal_seq_len=G, obs_seq_len=W, linear_output=linear, precision=prec).cuda()
        with torch.no_grad():
            for p in inner.parameters():
                p.add_(0.05 * torch.randn_like(p))
        model = GCDenoiser(inner, sigma_data=0.5).cuda().eval()
        lam = rng.choice([None, None, 0.0, 1.5]) if G > 0 else None
        call = model if lam is None else ClassifierFreeSampleModel(model, lam)
        state, action = torch.randn(B, t, obs, device="cuda"), torch.randn(B, t, act, device="cuda")
        goal = torch.randn(B, G, obs, device="cuda") if G > 0 else torch.zeros(B, 0, obs, device="cuda")
        sigma = torch.rand(B, device="cuda") * 0.9 + 0.05
        out = {}
        try:
            with torch.no_grad():
                for name, hint in (("small", _lib.PLAN_SMALL), ("per_op", _lib.PLAN_PER_OP)):
                    set_plan(forward=hint)
                    out[name] = call(state, action, goal, sigma)
        finally:
            set_plan(forward=0)
        desc = dict(D=D, H=H, W=W, G=G, obs=obs, act=act, L=L, B=B, t=t, prec=prec, linear=linear, lam=lam)
        assert torch.isfinite(out["small"]).all(), ("non-finite", desc)
        err = ((out["small"] - out["per_op"]).abs().max() / out["per_op"].abs().max().clamp_min(1e-6)).item()
        assert err < (2e-5 if prec == "fp32" else 3e-2), ("mismatch", err, desc)
        worst[prec] = max(worst[prec], err)
        n += 1
    print(f"fuzz_small: {n} random cases in {time.time() - t0:.0f} s; largest deviation of the small-batch path from the per-op kernels: "
          f"fp32 {worst['fp32']:.2e}, bf16 {worst['bf16']:.2e}")
    return n, worst


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
