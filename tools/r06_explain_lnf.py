#!/usr/bin/env python3
"""tools/fuzz_train_bf16.py seed 11 draws a block-push case of 12 token-window rows where ln_f.bias' gradient is
0.125 from the fp32 step in the library's plan and 0.016 in the per-op plan.  Which kernel, and is it rounding?
In BOTH plans ln_f.bias' gradient is produced by the same three launches (train.hip): tgemm_kernel (dxf = dpred W_head, fp32 out)
-> ln_bwd_kernel (block partials of sum_rows dxf) -> ln_reduce_kernel.  What differs between the plans is the FORWARD that
produced pred (train_fwd_kernel vs the per-op chain), hence dpred.  The sum is linear in dpred:
    grad(ln_f.bias) = sum_rows (dpred W) = (sum_rows dpred) W = grad(head bias) @ W_head      (MLP head: hidden bias grad @ W_hidden)
so each plan's ln_f.bias gradient can be re-evaluated in fp64 from THAT plan's own head-bias gradient and the bf16-rounded
head weight the kernel multiplied with.  If the kernels add up correctly the two agree to fp32 rounding, and the plan's
distance from the fp32 step is the distance of its dpred column sums -- twelve bf16-perturbed rows that cancel.
    python tools/r06_explain_lnf.py [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_train_bf16 as F  # noqa: E402


def bf16_round(x):
    return x.to(torch.bfloat16).double()


def on_case(desc, names, ref, lib, per_op, inner):
    if not (desc["D"] == 240 and desc["B"] * desc["t"] < 32):
        return False
    i_lnf = names.index("ln_f.bias")
    wname, bname = ("action_pred.weight", "action_pred.bias") if desc["linear"] else ("action_pred.0.weight", "action_pred.0.bias")
    i_w, i_b = names.index(wname), names.index(bname)
    W = dict(inner.named_parameters())[wname].detach()
    gmax = max(x.abs().max().item() for x in ref[1])
    floor = 2e-3 * gmax * ref[1][i_lnf].numel() ** 0.5
    d = lambda got: ((got[1][i_lnf] - ref[1][i_lnf]).norm() / max(ref[1][i_lnf].norm().item(), floor)).item()  # noqa: E731
    if max(d(lib), d(per_op)) < 0.05:
        return False
    print("case:", desc)
    print(f"loss: fp32 step {ref[0]:.6f}   library plan {lib[0]:.6f} ({abs(lib[0] - ref[0]) / abs(ref[0]):.2e})   per-op plan {per_op[0]:.6f} "
          f"({abs(per_op[0] - ref[0]) / abs(ref[0]):.2e})")
    print("head-bias gradient (= column sums of dpred over the", desc["B"] * desc["t"], "rows): fp32", [f"{v:+.5f}" for v in ref[1][i_b].tolist()],
          " library", [f"{v:+.5f}" for v in lib[1][i_b].tolist()], " per-op", [f"{v:+.5f}" for v in per_op[1][i_b].tolist()])
    print("head-weight gradient (no cancellation) vs fp32: library "
          f"{((lib[1][i_w] - ref[1][i_w]).norm() / ref[1][i_w].norm()).item():.4f}   per-op {((per_op[1][i_w] - ref[1][i_w]).norm() / ref[1][i_w].norm()).item():.4f}")
    print(f"|grad ln_f.bias| fp32 step {ref[1][i_lnf].norm().item():.4e}; the fuzz's floor for this tensor {floor:.4e}; "
          f"largest gradient entry of the step {gmax:.4e}")
    for name, got, Wm in (("fp32 step", ref, W.double()), ("library plan", lib, bf16_round(W)), ("per-op plan", per_op, bf16_round(W))):
        g_lnf, g_b = got[1][i_lnf].double(), got[1][i_b].double()
        re = g_b @ Wm                                       # fp64, from this plan's own dpred column sums and the weight it used
        print(f"{name:13s} distance of ln_f.bias from the fp32 step {d(got):.4f}   "
              f"kernel sum vs fp64 re-evaluation from its own head-bias gradient: {((g_lnf - re).norm() / g_lnf.norm()).item():.2e}   "
              f"head-bias gradient vs fp32 step: {((got[1][i_b] - ref[1][i_b]).norm() / ref[1][i_b].norm()).item():.4f}   "
              f"|head-bias gradient| {g_b.norm().item():.4e}")
    # what the same perturbation of dpred's column sums does through W: the distance is the head-bias distance, amplified by the
    # cancellation in (sum_rows dpred) W relative to the floor
    for name, got in (("library plan", lib), ("per-op plan", per_op)):
        delta = (got[1][i_b] - ref[1][i_b]).double() @ W.double()
        print(f"{name:13s} (head-bias gradient - fp32's) @ W_head: norm {delta.norm().item():.4e} = {delta.norm().item() / max(ref[1][i_lnf].norm().item(), floor):.4f} "
              f"of the yardstick: the whole distance is the forward's dpred, none of it the backward's summation")
    return True


if __name__ == "__main__":
    F.run(600.0, int(sys.argv[1]) if len(sys.argv) > 1 else 11, on_case=on_case)
