"""Parity and speed of the BF16X3 (split-bf16) instance of layers_kernel next to bf16 / fp32 (GPU box)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import beso_oracle as O
from conftest import load_golden, weights_from_fixture, rel_err
from test_gpu_parity import make_module, G, _weights, set_level

for fixture, cfg_name in [("kitchen_forward_std002.npz", "kitchen"), ("kitchen_forward_std008.npz", "kitchen"),
                          ("block_push_forward.npz", "block_push")]:
    fx = load_golden(fixture); cfg = O.CONFIGS[cfg_name]
    for prec in ("fp32", "bf16", "bf16x3"):
        m = make_module(cfg, _weights(fx, cfg), prec)
        worst = 0.0
        with torch.no_grad():
            for t in fx["ts"]:
                p = f"t{int(t)}::"
                s, a, g, sg = (G(fx[p + k]) for k in ("state", "action", "goal", "sigma"))
                e1 = rel_err(m(s, a, g, sg).cpu().numpy(), fx[p + "denoised"])
                e2 = rel_err(m(s, a, g, sg, uncond=True).cpu().numpy(), fx[p + "denoised_uncond"])
                e3 = rel_err(m.inner_model(s, a, g, sg).cpu().numpy(), fx[p + "inner"])
                worst = max(worst, e1, e2, e3)
        print(f"{fixture} {prec}: max rel err {worst:.3e}", flush=True)

from beso_amd import synthetic as S
cfg = S.SHAPES["kitchen"]; w = S.make_weights(cfg, seed=0, std=0.02)
for prec in ("bf16", "bf16x3"):
    m = make_module(O.CONFIGS["kitchen"], w, prec)
    for B in (64, 512, 4096):
        s_np, g_np, a_np = S.make_inputs(cfg, B, seed=1)
        s, g, a = (torch.from_numpy(v).cuda() for v in (s_np, g_np, a_np))
        sg = torch.full((B,), 0.3, device="cuda")
        with torch.no_grad():
            for _ in range(3): out = m(s, a, g, sg)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): out = m(s, a, g, sg)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"kitchen {prec} B={B}: {dt*1e3:.3f} ms/forward", flush=True)

# ---- sampler fixtures through the split-bf16 instance
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
fns = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun, "dpmpp_2m": ks.sample_dpmpp_2m,
       "dpm": ks.sample_dpm_2, "dpmpp_2s": ks.sample_dpmpp_2s}
for fixture, cfg_name in [("kitchen_samplers.npz", "kitchen"), ("block_push_heun_cfg.npz", "block_push")]:
    fx = load_golden(fixture); cfg = O.CONFIGS[cfg_name]
    for prec in ("fp32", "bf16x3"):
        m = make_module(cfg, _weights(fx, cfg), prec)
        lam = float(fx["cond_lambda"])
        model = m if lam < 0 else ClassifierFreeSampleModel(m, lam)
        for key in sorted(k[:-5] for k in fx if k.endswith("::out")):
            n = int(key.split("_")[-2]); sampler = key[: key.index(f"_{n}_")]
            out = fns[sampler](model, G(fx["state"]), G(fx["x_t"]), G(fx["goal"]), torch.from_numpy(fx[key + "::sigmas"]), disable=True)
            print(f"{fixture}:{key} {prec}: {rel_err(out.cpu().numpy(), fx[key + '::out']):.3e}", flush=True)

# ---- fused bf16 kernel against the per-op bf16 kernels (same arithmetic type, different rounding points)
from beso_amd import _lib
lib = _lib.load()
for cfg_name in ("kitchen", "block_push"):
    cfg = O.CONFIGS[cfg_name]
    for std in (0.02, 0.08):
        wts = O.make_weights(cfg, seed=3, std=std)
        m = make_module(cfg, wts, "bf16")
        s_np, g_np, a_np = O.make_inputs(cfg, 256, seed=9)
        sg_np = np.linspace(0.05, 1.0, 256).astype(np.float32)
        s, a, g, sg = G(s_np), G(a_np), G(g_np), G(sg_np)
        ref = O.denoise(wts, cfg, s_np, a_np, g_np, sg_np)
        outs = {}
        with torch.no_grad():
            for lvl in (2, 1, 0):
                set_level(lvl)
                outs[lvl] = m(s, a, g, sg).cpu().numpy()
        set_level(2)
        print(f"{cfg_name} std={std}: fused-vs-perop {rel_err(outs[2], outs[0]):.3e}  mlpblock-vs-perop {rel_err(outs[1], outs[0]):.3e}  "
              f"fused-vs-oracle {rel_err(outs[2], ref):.3e}  perop-vs-oracle {rel_err(outs[0], ref):.3e}", flush=True)
