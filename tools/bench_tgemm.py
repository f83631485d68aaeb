#!/usr/bin/env python3
"""Development aid: rate of the training GEMM (beso_debug_gemm, fp32 output) per operand layout and shape."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beso_amd import _lib  # noqa: E402


def run(lib, prec, aks, bks, M, N, K, splits, iters=20):
    dt = torch.float32 if prec == 1 else torch.bfloat16
    A = torch.randn((K, M) if aks else (M, K), device="cuda").to(dt)
    B = torch.randn((K, N) if bks else (N, K), device="cuda").to(dt)
    Cm = torch.zeros(M, N, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: _lib.check(lib.beso_debug_gemm(prec, aks, bks, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1],
                                                   Cm.data_ptr(), N, M, N, K, splits, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6


def main():
    lib = _lib.load_dev()
    T = 11264
    shapes = [("fwd qkv   NT", 0, 0, T, 1080, 360, 1), ("fwd fc1   NT", 0, 0, T, 1440, 360, 1), ("fwd fc2   NT", 0, 0, T, 360, 1440, 1),
              ("fwd proj  NT", 0, 0, T, 360, 360, 1), ("long-K    NT", 0, 0, T, 1440, 1440, 1), ("square    NT", 0, 0, 4096, 4096, 4096, 1),
              ("dgrad fc2 NK", 0, 1, T, 1440, 360, 1), ("dgrad fc1 NK", 0, 1, T, 360, 1440, 1), ("square    NK", 0, 1, 4096, 4096, 4096, 1),
              ("wgrad fc1 KK", 1, 1, 1440, 360, T, 15), ("wgrad fc2 KK", 1, 1, 360, 1440, T, 15), ("wgrad prj KK", 1, 1, 360, 360, T, 44),
              ("square    KK", 1, 1, 4096, 4096, 4096, 1)]
    for name, aks, bks, M, N, K, S in shapes:
        us, tf = run(lib, 0, aks, bks, M, N, K, S)
        print(f"{name}  M={M:6d} N={N:5d} K={K:6d} S={S:2d}: {us:8.1f} us  {tf:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
