#!/usr/bin/env python3
"""Does a hipGraph shorten the small-batch path's chain of launches?  The kitchen forward / 3-step DDIM at batch B, eager against
a captured graph replayed (GPU box):   python tools/r05_graph_small.py [B]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model
from beso_amd import synthetic as S
from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda:0"
cfg = S.SHAPES["kitchen"]
m = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), "bf16", dev)
s, g, a = (torch.from_numpy(v).to(dev) for v in S.make_inputs(cfg, B, seed=1))
sg = torch.full((B,), 0.3, device=dev)
sig = ks.get_sigmas_exponential(3, 0.005, 1.0)

def timed(fn, n=300):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

with torch.no_grad():
    for name, fn in (("forward", lambda: m(s, a, g, sg)), ("3-step DDIM", lambda: ks.sample_ddim(m, s, a, g, sig, disable=True))):
        eager = timed(fn)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fn()
        graphed = timed(gr.replay)
        print(f"kitchen bf16 B={B} {name}: eager {eager:7.1f} us   graph replay {graphed:7.1f} us", flush=True)
