#!/usr/bin/env python3
"""Replays tools/fuzz_train_bf16.py and prints, for every case whose bf16 loss is further from the fp32 step's than the tool's
bound for the library's plan, the same distance for the per-op plan (both are bf16 evaluations of the same sum):
    python tools/r06_loss_bound.py <seconds> <seed>"""
import importlib.util
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("fuzz_train_bf16", os.path.join(here, "fuzz_train_bf16.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
seen = [0, 0]


def on_case(desc, names, ref, a, b, inner):
    el = desc["B"] * desc["t"] * desc["act"]
    lb = max(3e-3, 3e-2 / (el ** 0.5 * min(0.5, abs(ref[0]) ** 0.5)))
    lb_old = max(3e-3, 6e-2 / el ** 0.5)
    le, le_op = abs(a[0] - ref[0]) / abs(ref[0]), abs(b[0] - ref[0]) / abs(ref[0])
    seen[0] += 1
    if le >= 0.7 * lb_old or le_op >= 0.7 * lb_old:
        seen[1] += 1
        print(f"case {seen[0]}: {el} loss elements, fp32 loss {ref[0]:.4e}: library {le:.3e}, per-op {le_op:.3e}, bound {lb:.3e} (the bound without the 1 / sqrt(loss) factor: {lb_old:.3e})  {desc}", flush=True)
    return False


try:
    mod.run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 23, on_case)
except AssertionError as e:
    print("assertion:", str(e)[:300])
print(f"{seen[0]} cases, {seen[1]} within 30 % of the loss bound or over it")
