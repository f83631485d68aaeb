import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from oracle import beso_oracle as O
from oracle import beso_oracle_torch as OT
from beso_amd import synthetic as S
cfg = S.SHAPES["kitchen"]; w = S.make_weights(cfg, seed=0, std=0.02)
ocfg = O.ScoreGPTConfig(**cfg.as_dict()); W = OT.to_torch(w)
for th in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    for B in (64, 1024, 4096):
        s, g, a = (torch.from_numpy(v) for v in S.make_inputs(cfg, B, seed=0)); sig = torch.full((B,), 0.3)
        OT.denoise(W, ocfg, s, a, g, sig)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 2.0:
            OT.denoise(W, ocfg, s, a, g, sig); n += 1
        dt = (time.perf_counter() - t0) / n
        print(f"threads {th:3d} B {B:4d}: {dt*1e3:8.1f} ms/forward {B/dt:9.0f} samples/s", flush=True)
