#!/usr/bin/env python3
"""Development aid: per-phase cycle breakdown of the training backward tile kernel (train_bwd_tail_kernel, workgroup 0) from
the in-kernel s_memtime stamps.  Needs a library built with -DBESO_FUSED_STAMPS=1 (tools/variants.py build st=-DBESO_FUSED_STAMPS=1;
BESO_HIP_LIB=beso_amd/lib/variants/libbeso_hip_st.so python tools/bwd_stamps.py [batch]).  Run on the GPU box."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from beso_amd import _lib              # noqa: E402
from beso_amd import synthetic as O    # noqa: E402

NAMES = {1: "start", 50: "stage q,k", 51: "bar", 52: "fetch v + gemm q", 53: "put v + gemm k + bars", 54: "gemm v | mlp->ln",
         55: "ln pass 1", 56: "ln bar 1", 57: "ln pass 2 (+stores)", 58: "ln frags + partials", 59: "ln bar 2",
         60: "h loads issued", 61: "fc1-like gemm", 62: "gelu' + dh stores", 63: "bar a", 64: "hT + colsum atomics", 65: "bar b",
         66: "fc2-like gemm", 67: "mlp end bar", 68: "proj gemm + dy stores"}


def main():
    from test_gpu_parity import _train_module, _train_inputs
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    cfg = O.SHAPES["kitchen"]
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.04), "bf16", attn_pdrop=0.3)
    state, action, goal, noise, sigma = _train_inputs(cfg, B, seed=5)
    step = m.hip_train_step(state, action, goal, noise, sigma)
    lib = _lib.load()
    lib.beso_debug_set_train_option(1, 1)
    buf = torch.zeros(8 * 2048, dtype=torch.int64, device="cuda:0")
    for _ in range(3):
        step.run(state, action, goal, noise, sigma, seed=1, fresh_grads=True)
    torch.cuda.synchronize()
    lib.beso_debug_set_stamps(buf.data_ptr(), buf.numel())
    step.run(state, action, goal, noise, sigma, seed=1, fresh_grads=True)
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
        if wave == 0 and 100 in real and 101 in real:
            dt_us = (real[101] - real[100]) / 100.0
            print(f"workgroup 0 lifetime {dt_us:.1f} us; shader clock {(ts[-1] - ts[0]) / dt_us:.0f} MHz")
        acc = collections.OrderedDict()
        for k in range(1, n):
            key = NAMES.get(int(ids[k]), str(ids[k]))
            acc[key] = acc.get(key, 0) + int(ts[k] - ts[k - 1])
        if keys is None:
            print(f"B={B}: {n} stamps/wave, wave 0 total {ts[-1] - ts[0]} cycles")
            keys = list(acc.keys())
            table = {k: [] for k in keys}
        for k in keys:
            table[k].append(acc.get(k, 0))
    print(f"  {'phase':26s} " + " ".join(f"   w{w}" for w in range(8)) + "   (kcycles per wave over the whole kernel)")
    for k in keys:
        print(f"  {k:26s} " + " ".join(f"{c / 1000:5.0f}" for c in table[k]))


if __name__ == "__main__":
    main()
