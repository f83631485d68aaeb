"""Hashes of short sampler calls on the small-batch path (kitchen, bf16 / fp32, B = 1, 3, 16, DDIM / Euler): run under two
libraries (BESO_HIP_LIB) and diff the output to see whether a change kept the results bit for bit (GPU box)."""
import sys, os, hashlib, torch
sys.path.insert(0, os.getcwd())
from bench import build_model
from beso_amd import synthetic as S
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
cfg = S.SHAPES["kitchen"]
out = []
for prec in ("bf16", "fp32"):
    m = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), prec, "cuda:0")
    for B in (1, 3, 16):
        s, g, a = (torch.from_numpy(v).to("cuda:0") for v in S.make_inputs(cfg, B, seed=1))
        for n in (3, 5):
            sig = ks.get_sigmas_exponential(n, 0.005, 1.0)
            with torch.no_grad():
                for name, fn in (("ddim", ks.sample_ddim), ("euler", ks.sample_euler)):
                    r = fn(m, s, a.clone(), g, sig, disable=True)
                    out.append(f"{prec} B={B} n={n} {name} {hashlib.sha1(r.cpu().numpy().tobytes()).hexdigest()[:12]} {float(r.abs().sum()):.6f}")
print("\n".join(out))
