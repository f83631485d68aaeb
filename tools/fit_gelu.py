#!/usr/bin/env python3
"""Fit of the transcendental-free GELU used by the fused MLP phase (beso_amd/csrc/fused.hip: gelu_fast):
Phi(v) - 1/2 = 0.5 erf(v/sqrt2) ~ vc * P(vc^2) on vc = clamp(v, +-vmax); iteratively re-weighted least
squares towards the minimax fit of the GELU error v * (approx - exact); evaluation checked in fp32."""
import numpy as np
from scipy.special import erf


def fit(deg=6, vmax=4.0):
    v = np.linspace(1e-4, vmax, 6000)
    s = v * v
    A = np.stack([v * s ** k for k in range(deg + 1)], 1)
    y = 0.5 * erf(v / np.sqrt(2))
    w = np.ones_like(v)
    for _ in range(80):
        c, *_ = np.linalg.lstsq(A * (w * v)[:, None], y * w * v, rcond=None)
        e = np.abs(v * (A @ c - y))
        w = w * (1 + 4 * e / e.max())
        w /= w.mean()
    return c


def max_error(c, vmax):
    vv = np.linspace(-10, 10, 400001).astype(np.float32)
    vc = np.clip(vv, -np.float32(vmax), np.float32(vmax))
    ss = (vc * vc).astype(np.float32)
    p = np.full_like(vv, np.float32(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        p = (p * ss + np.float32(c[k])).astype(np.float32)
    out = (vv * (np.float32(0.5) + (vc * p).astype(np.float32))).astype(np.float32)
    ref = 0.5 * vv.astype(np.float64) * (1 + erf(vv.astype(np.float64) / np.sqrt(2)))
    return float(np.abs(out - ref).max())


if __name__ == "__main__":
    c = fit()
    print("coefficients (low -> high order in vc^2):", [float(f"{x:.10g}") for x in c])
    print(f"max |GELU error| in fp32 over [-10, 10]: {max_error(c, 4.0):.3e}")
