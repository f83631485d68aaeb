#!/usr/bin/env python3
"""The BASELINE.json configurations other than the headline one (which is bench.py), measured on one
GPU through the public sampler API:  python tools/bench_configs.py [--out profiles/r01_configs.json]

  config 1  kitchen,      B = 64,   10-step DDIM            (the reference's own CPU-runnable case)
  config 4  block-push,   B = 2048, 50-step Heun (99 NFE), classifier-free guidance lambda = 2 (198 forwards / sample)
  config 5  long-horizon, B = 256 per GPU, 100-step Euler   (obs window 32, D = 512, H = 8, L = 6, G = 2: T = 1 + G + 64 = 67 tokens;
            the whole network is one launch per forward, a sample per workgroup.  The reference interleaves one action token
            per observation -- it has no action window apart from the obs window (score_gpts.py:330-331) -- so BASELINE's
            "action_window=8" is read as "of the 32 predicted actions keep the last 8": all 32 are computed and counted)

Reports wall time per sampler call, denoise-steps/s, NFE/s, sample*NFE/s and the achieved TFLOP/s
(algorithmic FLOPs per forward per sample x forwards).  Synthetic inputs, seeded weight recipe, bf16.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import synthetic as O  # noqa: E402   (synthetic-input and weight recipes only)
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel  # noqa: E402


def run(name, cfg, B, sampler, n_steps, smin, smax, lam=None, reps=5, precision="bf16"):
    dev = "cuda:0"
    model = build_model(cfg, O.make_weights(cfg, seed=0, std=0.02), precision, dev)
    call = model if lam is None else ClassifierFreeSampleModel(model, lam)
    s, g, a = (torch.from_numpy(v).to(dev) for v in O.make_inputs(cfg, B, seed=1))
    x_t = torch.randn_like(a) * smax
    sigmas = ks.get_sigmas_exponential(n_steps, smin, smax)
    fn = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun}[sampler]
    nfe = n_steps if sampler != "heun" else 2 * n_steps - 1
    fwd_per_nfe = 2 if (lam is not None and lam not in (0.0, 1.0)) else 1
    with torch.no_grad():
        fn(call, s, x_t, g, sigmas, disable=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn(call, s, x_t, g, sigmas, disable=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    assert torch.isfinite(out).all()
    flops = cfg.flops_per_sample() * B * nfe * fwd_per_nfe
    r = {"config": name, "batch": B, "sampler": sampler, "steps": n_steps, "nfe": nfe, "forwards_per_sample": nfe * fwd_per_nfe,
         "cond_lambda": lam, "precision": precision, "seconds_per_call": dt, "denoise_steps_per_s": n_steps / dt,
         "nfe_per_s": nfe / dt, "sample_nfe_per_s": B * nfe / dt, "tflops": flops / dt / 1e12,
         "frac_of_bf16_mfma_peak": flops / dt / 2.5e15}
    print(json.dumps(r), flush=True)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = [run("1: kitchen B=64 DDIM-10", O.SHAPES["kitchen"], 64, "ddim", 10, 0.005, 1.0, reps=20),
           run("4: block-push B=2048 Heun-50 CFG lambda=2", O.SHAPES["block_push"], 2048, "heun", 50, 0.05, 1.0, lam=2.0, reps=3),
           run("5: long-horizon B=256 Euler-100 (one GPU's shard)", O.SHAPES["long_horizon"], 256, "euler", 100, 0.005, 1.0, reps=2)]
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
