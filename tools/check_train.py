#!/usr/bin/env python3
"""Development aid: the HIP training step against torch autograd on the same function (GPU).
    python tools/check_train.py            # layout tests of the training GEMM + loss/grad parity (fp32, bf16)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beso_amd import _lib  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT  # noqa: E402
from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser  # noqa: E402


def gemm_case(lib, prec, aks, bks, M, N, K, splits):
    dev = "cuda"
    dt = torch.float32 if prec == 1 else torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + aks * 2 + bks)
    A = torch.randn((K, M) if aks else (M, K), generator=g).to(dev).to(dt)
    B = torch.randn((K, N) if bks else (N, K), generator=g).to(dev).to(dt)
    Cm = torch.zeros(M, N, device=dev)
    st = lib.beso_debug_gemm(prec, aks, bks, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], Cm.data_ptr(), N, M, N, K,
                             splits, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st, "debug_gemm")
    Am = (A.t() if aks else A).double()
    Bm = (B.t() if bks else B).double()
    ref = Am @ Bm.t()
    err = ((Cm.double() - ref).abs().max() / ref.abs().max()).item()
    return err


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autograd_reference import loss_autograd  # noqa: E402   (the comparator lives with the tests)


def main():
    lib = _lib.load_dev()
    worst = 0.0
    for prec in (1, 0):
        for aks, bks in ((0, 0), (0, 1), (1, 1)):
            for (M, N, K, S) in ((128, 128, 64, 1), (200, 136, 72, 1), (360, 1440, 1000, 3), (16, 360, 520, 2), (56, 360, 264, 1)):
                e = gemm_case(lib, prec, aks, bks, M, N, K, S)
                worst = max(worst, e)
                flag = "" if e < (2e-6 if prec == 1 else 2e-3) else "   <-- BAD"
                print(f"gemm prec={prec} aks={aks} bks={bks} M={M} N={N} K={K} S={S}: rel err {e:.2e}{flag}")
    torch.manual_seed(0)
    for name, kw, B in (("tiny", dict(state_dim=7, action_dim=3, embed_dim=48, n_layers=2, n_heads=6, goal_seq_len=2, obs_seq_len=3), 5),
                        ("kitchen", dict(state_dim=30, action_dim=9, embed_dim=360, n_layers=6, n_heads=6, goal_seq_len=2, obs_seq_len=4), 64)):
        for prec in ("fp32", "bf16"):
            inner = DiffusionGPT(device="cuda", goal_conditioned=True, embed_pdrob=0, attn_pdrop=0.0, resid_pdrop=0.0,
                                 linear_output=True, precision=prec, **kw).cuda()
            with torch.no_grad():
                for p in inner.parameters():
                    p.add_(0.05 * torch.randn_like(p))
            model = GCDenoiser(inner, sigma_data=0.5).cuda()
            model.train()
            t = kw["obs_seq_len"]
            state = torch.randn(B, t, kw["state_dim"], device="cuda")
            action = torch.randn(B, t, kw["action_dim"], device="cuda")
            goal = torch.randn(B, kw["goal_seq_len"], kw["state_dim"], device="cuda")
            noise = torch.randn_like(action)
            sigma = torch.rand(B, device="cuda") * 0.9 + 0.05
            loss_ref = loss_autograd(model, state, action, goal, noise.clone(), sigma)     # tests/autograd_reference.py
            loss_ref.backward()
            ref = [p.grad.clone() for p in inner.parameters()]
            for p in inner.parameters():
                p.grad = None
            loss = model.loss(state, action, goal, noise.clone(), sigma)
            assert loss.grad_fn is not None and "ScoreMatchingLoss" in type(loss.grad_fn).__name__, type(loss.grad_fn)
            loss.backward()
            torch.cuda.synchronize()
            print(f"{name} {prec}: loss hip {loss.item():.6f} ref {loss_ref.item():.6f}")
            worst_g = 0.0
            gmax = max(r.abs().max().item() for r in ref)
            for (n, p), r in zip(inner.named_parameters(), ref):
                # per tensor, relative to its own norm; tensors whose gradient is identically zero in exact
                # arithmetic (key.bias: softmax is shift invariant) are measured against the largest gradient entry
                e = ((p.grad - r).norm() / max(r.norm().item(), 1e-4 * gmax * r.numel() ** 0.5)).item()
                worst_g = max(worst_g, e)
                if e > (1e-3 if prec == "fp32" else 5e-2):
                    print(f"   {n}: rel err {e:.3e}  |ref| {r.norm().item():.3e}")
            print(f"   worst per-tensor gradient rel err {worst_g:.3e}")


if __name__ == "__main__":
    main()
