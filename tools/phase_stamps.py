#!/usr/bin/env python3
"""Development aid: per-phase cycle breakdown of the fused layers kernel (workgroup 0, wave 0), from
the in-kernel s_memtime stamps (beso_debug_set_stamps).  Run on the GPU box."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model          # noqa: E402
from beso_amd import _lib              # noqa: E402
from beso_amd import synthetic as O    # noqa: E402

NAMES = {40: "emb_prologue", 41: "emb_loads_issued", 42: "emb_bias", 43: "emb_mfma", 1: "start", 2: "layer_start", 7: "ln1_done", 10: "pair_start(proj B)", 11: "qkv_gemm(pair)+write(A)", 12: "bar_qkv",
         13: "proj(A)+write(B)", 14: "bar_B", 15: "core(B)", 18: "bar_coreB", 16: "core(A)", 17: "bar_coreA", 3: "attn_done(proj B)",
         6: "ln2_done", 20: "fc1(0)", 21: "fc2(c-1)||gelu(c)", 22: "bar_a", 23: "hT_write+fc1(c+1)", 24: "bar_b", 25: "fc2(last)", 4: "layers_done",
         5: "stored", 30: "ln_pass1", 31: "ln_bar1", 32: "ln_pass2", 33: "ln_bar2", 34: "ln_write"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = "cuda:0"
    cfg = O.SHAPES[sys.argv[2] if len(sys.argv) > 2 else "kitchen"]
    model = build_model(cfg, O.make_weights(cfg, seed=0, std=0.02), "bf16", dev)
    if len(sys.argv) > 3:          # classifier-free guidance: python tools/phase_stamps.py 2048 block_push 2.0
        from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
        model = ClassifierFreeSampleModel(model, cond_lambda=float(sys.argv[3]))
    s, g, a = (torch.from_numpy(v).to(dev) for v in O.make_inputs(cfg, B, seed=1))
    sig = torch.full((B,), 0.3, device=dev)
    # the stamps exist in a development build with the stamp code compiled in -- the library the model itself runs on:
    #   python tools/variants.py build st=-DBESO_DEV_API=1,-DBESO_FUSED_STAMPS=1
    #   BESO_HIP_LIB=beso_amd/lib/variants/libbeso_hip_st.so python tools/phase_stamps.py
    import ctypes as C
    lib = _lib.load()
    if not hasattr(lib, "beso_debug_set_stamps"):
        raise SystemExit("run with BESO_HIP_LIB=<a -DBESO_DEV_API=1 -DBESO_FUSED_STAMPS=1 build> (see the comment above)")
    lib.beso_debug_set_stamps.restype = None
    lib.beso_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
    buf = torch.zeros(8 * 2048, dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model(s, a, g, sig)
        torch.cuda.synchronize()
        lib.beso_debug_set_stamps(buf.data_ptr(), buf.numel())
        model(s, a, g, sig)
        torch.cuda.synchronize()
        lib.beso_debug_set_stamps(None, 0)
    allv = buf.cpu().numpy().reshape(8, -1)
    t0 = None
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
        if wave == 0 and 100 in real and 101 in real:
            dt_us = (real[101] - real[100]) / 100.0
            print(f"workgroup 0 lifetime {dt_us:.1f} us; shader clock {(ts[-1] - ts[0]) / dt_us:.0f} MHz")
        total = ts[-1] - ts[0]
        acc = collections.OrderedDict()
        for k in range(1, n):
            key = NAMES.get(int(ids[k]), str(ids[k]))
            acc[key] = acc.get(key, 0) + int(ts[k] - ts[k - 1])
        if wave == 0:
            print(f"B={B}: {n} stamps/wave, wave 0 total {total} cycles")
            keys = list(acc.keys())
            table = {k: [] for k in keys}
        for k in keys:
            table[k].append(acc.get(k, 0))
    print(f"  {'phase':22s} " + " ".join(f"   w{w}" for w in range(8)) + "   (kcycles per wave over the whole kernel)")
    for k in keys:
        print(f"  {k:22s} " + " ".join(f"{c / 1000:5.0f}" for c in table[k]))


if __name__ == "__main__":
    main()
