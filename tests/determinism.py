"""Run-to-run stability of every instance of the one-launch kernel on the MI355X: each (config, precision, batch, window)
is evaluated REPS times -- one forward, one classifier-free forward where the shape has it, one sampler loop -- and every
repetition must equal the first one bit for bit; the first one is also compared with the block / per-op kernels of the
same library (a nondeterministic instance shows up as either).  Usage: python tests/determinism.py [--reps 16] [--quick]
Exit code 1 on any difference.  (tests/test_gpu_parity.py::test_one_launch_kernels_are_stable_from_run_to_run runs the
--quick set.)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

CASES = {   # config -> precision -> [(B, t)]   (t = None: the config's full window)
    "long_horizon": {"bf16": [(1, None), (3, None), (37, 16), (2, 31), (256, None)], "fp16": [(64, None)],
                     "bf16x3": [(8, None)]},
    "kitchen": {"bf16": [(2, None), (513, None), (4096, None)], "fp16": [(4096, None)], "bf16x3": [(512, None), (4096, None)]},
    "block_push": {"bf16": [(7, None), (2048, None)], "fp16": [(2048, None)], "bf16x3": [(2048, None)]},
}
QUICK = {
    "long_horizon": {"bf16": [(3, None), (64, None), (2, 31)]},
    "kitchen": {"bf16x3": [(4096, None)], "bf16": [(4096, None)]},
}


def run(cases, reps, verbose=True):
    from oracle import beso_oracle as O
    from conftest import rel_err
    from test_gpu_parity import make_module, G, set_level
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    bad = []
    for cfg_name, by_prec in cases.items():
        cfg = O.CONFIGS[cfg_name]
        wts = O.make_weights(cfg, seed=3, std=0.03)
        for prec, shapes in by_prec.items():
            m = make_module(cfg, wts, prec)
            cfgm = ClassifierFreeSampleModel(m, 1.5)
            sig = ks.get_sigmas_exponential(4, 0.05, 1.0)
            for B, t in shapes:
                t = t or cfg.obs_seq_len
                s, g, a = (G(v) for v in O.make_inputs(cfg, B, seed=B + t, t=t))
                sg = G(np.linspace(0.05, 1.0, B).astype(np.float32))
                fns = {"forward": lambda: m(s, a, g, sg), "euler": lambda: ks.sample_euler(m, s, a, g, sig, disable=True)}
                if cfg_name != "long_horizon":
                    fns["cfg"] = lambda: cfgm(s, a, g, sg)
                    fns["heun_cfg"] = lambda: ks.sample_heun(cfgm, s, a, g, sig, disable=True)
                with torch.no_grad():
                    for name, fn in fns.items():
                        first = fn()
                        diffs = [float((fn() - first).abs().max()) for _ in range(reps)]
                        other = None
                        if prec == "bf16":
                            try:
                                set_level(1 if cfg_name == "long_horizon" else 0)
                                other = rel_err(first.cpu().numpy(), fn().cpu().numpy())
                            finally:
                                set_level(2)
                        ok = torch.isfinite(first).all().item() and max(diffs) == 0.0
                        if verbose or not ok:
                            print(f"[determinism] {cfg_name:12s} {prec:6s} B={B:<5d} t={t:<3d} {name:9s} run-to-run max |diff| "
                                  f"{max(diffs):.3e} over {reps} reps" + (f"; vs the other kernels {other:.3e}" if other is not None else "")
                                  + ("" if ok else "   <-- UNSTABLE"), flush=True)
                        if not ok:
                            bad.append((cfg_name, prec, B, t, name, max(diffs)))
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=16)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    bad = run(QUICK if a.quick else CASES, a.reps)
    print("[determinism]", "all stable" if not bad else f"{len(bad)} unstable cases: {bad}")
    sys.exit(1 if bad else 0)
