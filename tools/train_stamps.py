#!/usr/bin/env python3
"""Development aid: per-phase cycle breakdown of the training step's data-gradient kernels (workgroup 0, every wave), from
in-kernel s_memtime stamps.  Every launch of a stamped kernel starts at the head of the buffer, so what is read back is the
LAST launch of the step: layer 0's train_mlp_bwd_kernel (kind 2) or layer 0's q|k|v train_dgrad_kernel (kind 3).
    python tools/variants.py build st2=-DBESO_DEV_API=1,-DBESO_FUSED_STAMPS=2 st3=-DBESO_DEV_API=1,-DBESO_FUSED_STAMPS=3
    BESO_HIP_LIB=beso_amd/lib/variants/libbeso_hip_st2.so python tools/train_stamps.py 8192 [kitchen|block_push]"""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import _lib  # noqa: E402
from beso_amd import synthetic as O  # noqa: E402
from _agent import build_agent  # noqa: E402
from beso_amd.networks.scaler.scaler_class import Scaler  # noqa: E402

NAMES = {50: "start", 51: "stage dyo tile (loads + LDS)", 52: "barrier", 53: "FC2 dgrad GEMM (h loads, ring start)",
         54: "GELU' epilogue + dh store", 55: "barrier (dh chunk)", 56: "FC1 dgrad GEMM (ring start)",
         57: "LN: partial sums out", 58: "barrier", 59: "proj dgrad GEMM (ring start)", 60: "dy store",
         61: "LN: stats / x / dres loads -> xh", 62: "LN: barrier 1", 63: "LN: row sums -> LDS", 64: "LN: barrier 2",
         65: "LN: dres / dxb out (+ emit)",
         70: "start", 71: "stage dqkv tile (loads + LDS)", 72: "barrier", 73: "q|k|v dgrad GEMM (ring start)",
         74: "LN: partial sums out / plain store"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    name = sys.argv[2] if len(sys.argv) > 2 else "kitchen"
    dev = "cuda:0"
    cfg = O.SHAPES[name]
    w = O.make_weights(cfg, seed=0, std=0.02)
    attn_p, resid_p = {"kitchen": (0.3, 0.0), "block_push": (0.05, 0.05)}.get(name, (0.0, 0.0))

    def model():
        m = build_model(cfg, w, "bf16", dev)
        inner = m.inner_model
        inner._pdrops = (0.0, attn_p, resid_p)
        for blk in inner.blocks:
            blk.attn.attn_drop.p = attn_p
            blk.attn.resid_drop.p = resid_p
            for mod in blk.mlp:
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = resid_p
        return m

    agent = build_agent(cfg, model, device=dev)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((256, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((256, cfg.act_dim)).astype(np.float32), True, dev))
    agent.set_bounds(agent.scaler)
    torch.manual_seed(1234)
    batch = {"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim), "action": torch.randn(B, cfg.obs_seq_len, cfg.act_dim),
             "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim)}
    batch = {k: v.to(dev) for k, v in batch.items()}
    lib = _lib.load()
    if not hasattr(lib, "beso_debug_set_stamps"):
        raise SystemExit("run with BESO_HIP_LIB=<a -DBESO_DEV_API=1 -DBESO_FUSED_STAMPS=2|3 build> (see the docstring)")
    lib.beso_debug_set_stamps.restype = None
    lib.beso_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
    buf = torch.zeros(8 * 2048, dtype=torch.int64, device=dev)
    for _ in range(5):
        agent.train_step(batch)
    torch.cuda.synchronize()
    lib.beso_debug_set_stamps(buf.data_ptr(), buf.numel())
    agent.train_step(batch)
    torch.cuda.synchronize()
    lib.beso_debug_set_stamps(None, 0)
    allv = buf.cpu().numpy().reshape(8, -1)
    keys, table = None, None
    for wave in range(8):
        v = allv[wave]
        ids, ts = v[0::2], v[1::2]
        n = int(np.nonzero(ids)[0].max()) + 1 if ids.any() else 0
        ids, ts = ids[:n], ts[:n]
        if n == 0:
            continue
        real = {int(i): int(t) for i, t in zip(ids, ts) if i >= 100}
        keep = ids < 100
        ids, ts = ids[keep], ts[keep]
        n = len(ids)
        total = int(ts[-1] - ts[0])
        if wave == 0:
            if 100 in real and 101 in real:
                dt_us = (real[101] - real[100]) / 100.0
                print(f"{name} B={B}: workgroup 0 lifetime {dt_us:.1f} us; shader clock {total / max(dt_us, 1e-9):.0f} MHz")
            print(f"{n} stamps/wave, wave 0 total {total} cycles")
        acc = collections.OrderedDict()
        for k in range(1, n):
            key = NAMES.get(int(ids[k]), str(ids[k]))
            acc[key] = acc.get(key, 0) + int(ts[k] - ts[k - 1])
        if keys is None:
            keys = list(acc.keys())
            table = {k: [] for k in keys}
            tot = []
        for k in keys:
            table[k].append(acc.get(k, 0))
        tot.append(total)
    if keys is None:
        raise SystemExit("no stamps recorded: is this a -DBESO_FUSED_STAMPS=2|3 build?")
    print(f"  {'phase (cycles up to this stamp)':40s} " + " ".join(f"    w{w}" for w in range(len(tot))) + "   share(w0)")
    for k in keys:
        print(f"  {k:40s} " + " ".join(f"{c / 1000:6.1f}" for c in table[k]) + f"   {100.0 * table[k][0] / tot[0]:5.1f} %")
    print(f"  {'total (kcycles)':40s} " + " ".join(f"{c / 1000:6.1f}" for c in tot))


if __name__ == "__main__":
    main()
