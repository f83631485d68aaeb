#!/usr/bin/env python3
"""Development aid (GPU box): random (shape, batch, window, lambda) forwards through the one-launch kernels against the
per-op / block-kernel form of the same library (beso_debug_set_fused_level_max) -- bf16, both carry their own rounding, so
the bound is the test suite's: relative difference < 2e-2 of the output's max (4e-2 under classifier-free guidance) -- and,
for small batches, against the oracle.  (A miss is examined with the BF16X3 / fp32 modes of the same kernels: 1e-5 / 2e-6
there means rounding, not a defect.)   python tests/fuzz_forward.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import test_gpu_parity as T
    from beso_amd import _lib
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    O = T.O
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _lib.load()
    mods = {}
    worst = {}
    for case in range(n_cases):
        name = ["kitchen", "block_push", "long_horizon"][int(rng.integers(0, 3))]
        cfg = O.CONFIGS[name]
        if name not in mods:
            w = O.make_weights(cfg, seed=11, std=0.03)
            mods[name] = (w, T.make_module(cfg, w, "bf16"))
        w, m = mods[name]
        B = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 64, 65, 200, 513, int(rng.integers(1, 700))]))
        if name == "long_horizon":
            B = min(B, 96)
        t = int(rng.integers(1, cfg.obs_seq_len + 1))
        lam = float(rng.choice([1.0, 1.0, 2.0, 0.0]))
        s_np, g_np, a_np = O.make_inputs(cfg, B, seed=case, t=t)
        sg_np = np.exp(rng.uniform(np.log(0.02), np.log(1.0), B)).astype(np.float32)       # the shipped sigma range, log-uniform
        model = m if lam == 1.0 else ClassifierFreeSampleModel(m, lam)
        outs = {}
        try:
            with torch.no_grad():
                for lvl in (2, 0):
                    T.set_level(lvl)
                    outs[lvl] = model(T.G(s_np), T.G(a_np), T.G(g_np), T.G(sg_np)).cpu().numpy()
        finally:
            T.set_level(2)
        e = T.rel_err(outs[2], outs[0])
        eo = -1.0
        if B <= 9:
            eo = T.rel_err(outs[2], O.denoise_cfg(w, cfg, s_np, a_np, g_np, sg_np, lam))
        # classifier-free pairs amplify both members' rounding by (1 + 2 lambda); a handful of outputs (B = 1) has no averaging
        bound = 2e-2 * (1.0 if lam in (0.0, 1.0) else 2.0)
        ok = np.isfinite(outs[2]).all() and e < bound and eo < bound
        key = name
        worst[key] = max(worst.get(key, 0.0), e)
        print(f"{case:3d} {name:13s} B={B:4d} t={t:2d} lam={lam}: fused-vs-per-op {e:.2e} vs-oracle {eo:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            sys.exit(1)
    print("worst fused-vs-per-op per shape:", {k: f"{v:.2e}" for k, v in worst.items()})


if __name__ == "__main__":
    main()
