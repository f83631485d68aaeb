#!/usr/bin/env python3
"""Random model shapes through the bf16 HIP training step: the library's kernels and the per-op plan, each against the fp32 HIP step
of the same weights (same seed = same dropout masks); the shipped shapes at random batch sizes / windows / dropouts among them:
    python tools/fuzz_train_bf16.py [seconds] [seed]
What it exercises beyond the parity tests: the matrix-pipe attention backward (any T <= 16, hd <= 64, hd % 4 == 0), the row ranges
of the grouped weight-gradient launch (small models), ragged last workgroups of the data-gradient kernels, dropout in the
LayerNorm-backward epilogue.  Bound: the parity tests' 2.6e-2 per tensor (or 1.5 x the per-op plan's distance), loss 3e-3
(6e-2 / sqrt(elements) for tiny batches; gradients of batches under 32 rows: 0.15)."""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from beso_amd import _lib  # noqa: E402
from beso_amd.runtime import set_plan  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser  # noqa: E402


def run(budget=60.0, seed=0, on_case=None):
    """on_case(desc, names, ref, lib_plan, per_op_plan, inner): called for every case before its assertion; a true return value
    ends the run (tools/r06_explain_lnf.py replays a seed up to the case it dissects)."""
    rng = random.Random(seed)
    torch.manual_seed(rng.randrange(1 << 30))
    t0, n, worst, worst_at, worst_ratio = time.time(), 0, 0.0, None, 0.0
    shipped = [dict(D=360, H=6, W=4, G=2, obs=30, act=9, L=6), dict(D=240, H=12, W=5, G=1, obs=10, act=2, L=4)]      # kitchen, block-push
    while time.time() - t0 < budget:
        if rng.random() < 0.3:
            c = dict(rng.choice(shipped)); c["L"] = rng.randint(1, c["L"])
            D, H, W, G, obs, act, L = c["D"], c["H"], c["W"], c["G"], c["obs"], c["act"], c["L"]
            B = rng.choice([1, 3, 4, 5, 37, 100, 257, 1025])
        else:
            H = rng.choice([1, 2, 3, 4, 6, 8])
            hd = rng.choice([4, 8, 12, 16, 20, 32, 40, 60, 64])
            D = H * hd
            if D % 8 or D > 512:
                continue
            W, G = rng.randint(1, 7), rng.randint(0, 2)
            obs, act, L, B = rng.randint(1, 40), rng.randint(1, 12), rng.randint(1, 3), rng.randint(1, 70)
        t = rng.randint(1, W)
        attn_p = rng.choice([0.0, 0.3]); resid_p = rng.choice([0.0, 0.0, 0.1]); embed_p = rng.choice([0.0, 0.0, 0.1])
        linear = rng.random() < 0.7
        inner = DiffusionGPT(state_dim=obs, device="cuda", goal_conditioned=G > 0, action_dim=act, embed_dim=D, embed_pdrob=embed_p,
                             attn_pdrop=attn_p, resid_pdrop=resid_p, n_layers=L, n_heads=H, goal_seq_len=G, obs_seq_len=W,
                             linear_output=linear, precision="bf16").cuda()
        with torch.no_grad():
            for p in inner.parameters():
                p.add_(0.05 * torch.randn_like(p))
        model = GCDenoiser(inner, sigma_data=0.5).cuda().train()
        state, action = torch.randn(B, t, obs, device="cuda"), torch.randn(B, t, act, device="cuda")
        goal = torch.randn(B, G, obs, device="cuda") if G > 0 else None
        noise, sigma = torch.randn_like(action), torch.rand(B, device="cuda") * 0.9 + 0.05
        # the fp32 HIP step of the same weights (3.5e-5 from torch autograd: tools/fuzz_train.py) with the same seed = the same
        # dropout masks is the yardstick: a bf16 plan is measured by its distance from it
        inner32 = DiffusionGPT(state_dim=obs, device="cuda", goal_conditioned=G > 0, action_dim=act, embed_dim=D, embed_pdrob=embed_p,
                               attn_pdrop=attn_p, resid_pdrop=resid_p, n_layers=L, n_heads=H, goal_seq_len=G, obs_seq_len=W,
                               linear_output=linear, precision="fp32").cuda()
        inner32.load_state_dict(inner.state_dict())
        model32 = GCDenoiser(inner32, sigma_data=0.5).cuda().train()
        step, step32 = model.hip_train_step(state, action, goal, noise, sigma), model32.hip_train_step(state, action, goal, noise, sigma)
        assert step is not None and step32 is not None
        seed = 1234 + n
        r = step32.run(state, action, goal, noise, sigma, seed=seed, fresh_grads=True)
        ref = (r[0].item(), [v.clone() for v in r[2]])
        out = {}
        try:
            for plan in (0, _lib.TRAIN_PLAN_PER_OP):
                set_plan(train=plan)
                r = step.run(state, action, goal, noise, sigma, seed=seed, fresh_grads=True)
                out[plan] = (r[0].item(), [v.clone() for v in r[2]])
        finally:
            set_plan(train=0)
        a, b = out[0], out[_lib.TRAIN_PLAN_PER_OP]
        gmax = max(x.abs().max().item() for x in ref[1])
        # Every tensor is measured, against max(its own norm, 2e-3 x the largest gradient entry x sqrt(elements)): key.bias, whose
        # exact gradient is zero (softmax is invariant to a shift of a row's scores: what a plan holds there is the rounding noise
        # of the summands), and the handful-of-elements tensors (the output bias: each element a sum over all B t rows with
        # cancellation) are thereby held to the size of a rounding error of the LARGE gradients instead of being left out
        names = [k for k, _ in inner.named_parameters()]
        def dist(got, who=False):
            # (key.bias: twice the floor -- nothing but noise is measured there, and its size varies by a factor of two
            #  between the plans at batches of one or two samples.  Tensors of fewer than 16 elements -- the output bias: each
            #  element ONE sum over all B t rows of bf16-rounded summands that cancel -- are held to 1e-1 x the largest gradient
            #  entry per element: the bound then allows an absolute error of 2.6e-3 of it, one bf16 rounding of a number of
            #  that size; both plans measure 1e-3 ... 3e-3 there at the fuzz's batch sizes, see profiles/README.md)
            def floor(k, y):
                f = 4e-3 if k.endswith("attn.key.bias") else (1e-1 if y.numel() < 16 else 2e-3)
                return f * gmax * y.numel() ** 0.5
            d = [(((x - y).norm() / max(y.norm().item(), floor(k, y), 1e-12)).item(), k) for k, x, y in zip(names, got[1], ref[1])]
            return max(d) if who else max(d)[0]
        e, e_op = dist(a), dist(b)
        le = abs(a[0] - ref[0]) / abs(ref[0])
        desc = dict(D=D, H=H, W=W, G=G, obs=obs, act=act, L=L, B=B, t=t, linear=linear, attn_p=attn_p, resid_p=resid_p, embed_p=embed_p)
        if on_case is not None and on_case(desc, names, ref, a, b, inner):
            return n, worst, worst_ratio
        assert all(torch.isfinite(x).all() for x in a[1]), ("non-finite gradient", desc)
        # (a loss over a handful of elements does not average the bf16 rounding: two elements measured 3.0e-2 in round 5's seed 11.
        #  And a SMALL loss amplifies it: the loss is mean(err^2), a forward deviation d of the prediction moves it by 2 d / err
        #  relative -- round 6's seed 23 draws B t act = 2 with an fp32 loss of 0.031 (err ~ 0.18): the library's plan lands
        #  6.2e-2 away, the per-op plan 8.9e-3, i.e. forward deviations of 5e-3 and 8e-4 of a prediction of size 1, both inside
        #  the forward's bf16 bound (profiles/r06_loss_bound.txt).  The bound is the old one for losses >= 0.25 and grows as
        #  1 / sqrt(loss) below.)
        lb = max(3e-3, 3e-2 / ((B * t * act) ** 0.5 * min(0.5, abs(ref[0]) ** 0.5)))
        # the library's kernels may not be further from fp32 than the bf16 bound of the parity tests, or -- tiny batches, where
        # every bf16 evaluation is that far off -- than 1.5 x the per-op plan's own distance
        # (batches of fewer than 32 token-window rows: the bias-like gradients -- ln_f.bias, the output bias -- are sums over a
        #  dozen rows that cancel, and which of two bf16 evaluations lands closer to fp32 is luck: round 5's seed 11 draws a
        #  block-push case with B t = 12 where the library measures 0.125 and the per-op plan 0.016 on ln_f.bias -- with the
        #  round-4 library to the last digit, and 9e-3 / 1e-2 over twelve other seeds of the same shape.  Such batches are held
        #  to 0.15: an indexing or masking error shows as O(1).)
        cap = 2.6e-2 if B * t >= 32 else 0.15
        assert e < max(cap, 1.5 * e_op) and le < lb, ("mismatch", dist(a, True), dist(b, True), le, desc)
        if e > worst:
            worst, worst_at = e, (dist(a, True)[1], round(e_op, 4), desc)
        worst_ratio = max(worst_ratio, e / max(e_op, 1e-3))
        n += 1
    print(f"fuzz_train_bf16: {n} random cases in {time.time() - t0:.0f} s, largest distance of the library's plan from the fp32 step {worst:.2e} "
          f"(tensor, the per-op plan's distance, case: {worst_at}); largest library / per-op distance ratio {worst_ratio:.2f}")


    return n, worst, worst_ratio


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
