#!/usr/bin/env python3
"""Latency of one GCDenoiser.forward (fused bf16 path) vs batch size through the two instances of the fused kernel:
the throughput instance (8 samples per workgroup) and the latency instance (2 per workgroup), selected with
the BESO_PLAN_SPW8 hint.   python tools/latency_instances.py [kitchen|block_push]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import _lib, synthetic as S  # noqa: E402
from beso_amd.runtime import set_plan  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
    dev = "cuda:0"
    cfg = S.SHAPES[name]
    model = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), "bf16", dev)
    lib = _lib.load()
    for B in (1, 2, 8, 32, 64, 128, 256, 384, 512, 768, 1024, 2048):
        s, g, a = (torch.from_numpy(v).to(dev) for v in S.make_inputs(cfg, B, seed=1))
        sg = torch.full((B,), 0.3, device=dev)
        row, outs = [], []
        for limit in (0, 512):
            set_plan(forward=_lib.PLAN_SPW8 if limit == 0 else 0)
            with torch.no_grad():
                for _ in range(5):
                    out = model(s, a, g, sg)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 100
                for _ in range(n):
                    model(s, a, g, sg)
                torch.cuda.synchronize()
                row.append((time.perf_counter() - t0) / n * 1e6)
                outs.append(out.clone())
        same = torch.equal(outs[0], outs[1])
        print(f"{name} B={B:5d}  8 per workgroup {row[0]:8.1f} us   latency instances (2 / 4 per workgroup) {row[1]:8.1f} us   bit-identical {same}", flush=True)


if __name__ == "__main__":
    main()
