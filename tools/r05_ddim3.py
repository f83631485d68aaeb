import sys, os, torch
sys.path.insert(0, os.getcwd())
from bench import build_model
from beso_amd import synthetic as S
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
prec = sys.argv[1]
cfg = S.SHAPES["kitchen"]
m = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), prec, "cuda:0")
sig = ks.get_sigmas_exponential(3, 0.005, 1.0)
for B in [int(b) for b in sys.argv[2:]]:
    s, g, a = (torch.from_numpy(v).to("cuda:0") for v in S.make_inputs(cfg, B, seed=1))
    with torch.no_grad():
        for _ in range(30): ks.sample_ddim(m, s, a, g, sig, disable=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): ks.sample_ddim(m, s, a, g, sig, disable=True)
        e1.record(); torch.cuda.synchronize()
    print(prec, B, round(e0.elapsed_time(e1) * 10, 1), "us per 3-step DDIM", flush=True)
