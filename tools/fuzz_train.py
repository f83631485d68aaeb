#!/usr/bin/env python3
"""Random model shapes through the HIP training step (fp32 mode) against torch autograd on the same function:
    python tools/fuzz_train.py [seconds] [seed]
Shapes: embed_dim = 8 * heads * k, 1-4 layers, window 1-9, goal length 0-3, obs / act dims 1-40 / 1-12, batch 1-70."""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser  # noqa: E402


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autograd_reference import loss_autograd  # noqa: E402   (the comparator lives with the tests)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.manual_seed(rng.randrange(1 << 30))
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < budget:
        H = rng.choice([1, 2, 3, 4, 6, 8])
        hd = rng.choice([4, 8, 12, 16, 20, 30, 40, 60, 64])
        D = H * hd
        if D % 8 or D > 512:
            continue
        W, G = rng.randint(1, 9), rng.randint(0, 3)
        obs, act, L, B = rng.randint(1, 40), rng.randint(1, 12), rng.randint(1, 4), rng.randint(1, 70)
        t = rng.randint(1, W)
        linear = rng.random() < 0.7
        inner = DiffusionGPT(state_dim=obs, device="cuda", goal_conditioned=G > 0, action_dim=act, embed_dim=D, embed_pdrob=0,
                             attn_pdrop=0.0, resid_pdrop=0.0, n_layers=L, n_heads=H, goal_seq_len=G, obs_seq_len=W,
                             linear_output=linear, precision="fp32").cuda()
        with torch.no_grad():
            for p in inner.parameters():
                p.add_(0.05 * torch.randn_like(p))
        model = GCDenoiser(inner, sigma_data=0.5).cuda().train()
        state, action = torch.randn(B, t, obs, device="cuda"), torch.randn(B, t, act, device="cuda")
        goal = torch.randn(B, G, obs, device="cuda") if G > 0 else None
        noise, sigma = torch.randn_like(action), torch.rand(B, device="cuda") * 0.9 + 0.05
        ref_loss = loss_autograd(model, state, action, goal, noise.clone(), sigma)     # tests/autograd_reference.py
        ref_loss.backward()
        ref = [p.grad.clone() for p in inner.parameters()]
        for p in inner.parameters():
            p.grad = None
        loss = model.loss(state, action, goal, noise.clone(), sigma)
        assert "ScoreMatchingLoss" in type(loss.grad_fn).__name__, (D, H, W, G, obs, act, L, B, t)
        loss.backward()
        gmax = max(r.abs().max().item() for r in ref)
        errs = [((p.grad - r).norm() / max(r.norm().item(), 1e-4 * gmax * r.numel() ** 0.5)).item()
                for p, r in zip(inner.parameters(), ref)]
        e = max(max(errs), abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()))
        worst = max(worst, e)
        assert e < 5e-4, ("mismatch", e, dict(D=D, H=H, W=W, G=G, obs=obs, act=act, L=L, B=B, t=t, linear=linear))
        n += 1
    print(f"fuzz_train: {n} random shapes in {time.time() - t0:.0f} s, worst error vs autograd {worst:.2e}")


if __name__ == "__main__":
    main()
