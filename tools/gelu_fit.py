#!/usr/bin/env python3
"""The polynomial of csrc/mmhip_common.h: mm_gelu.  GELU(x) = max(x, 0) - |x| Phi(-|x|); Phi(-a) = 2^-(1 + P(a)), P of degree `--deg` fitted on
Chebyshev nodes of [0, X] to -log2(2 Phi(-a)) (scipy log_ndtr, double), converted to the power basis, rounded to fp32; the check evaluates the
kernel's own arithmetic in fp32 (Horner, exp2) on 2.4 M points of [-12, 12] against x erfc(-x / sqrt 2) / 2 in double.  CPU only.
    python tools/gelu_fit.py [--deg 8] [--X 5.5]"""
import argparse

import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as Pl
from scipy.special import erfc, log_ndtr


def fit(deg, X):
    k = np.arange(4 * deg)
    nodes = np.cos(np.pi * (k + 0.5) / (4 * deg))
    a = (nodes + 1) * X / 2
    c = Ch.chebfit(nodes, -(log_ndtr(-a) / np.log(2) + 1), deg)
    p, lin, pu = Ch.cheb2poly(c), np.array([-1.0, 2.0 / X]), np.zeros(1)
    for i, ci in enumerate(p):
        pu = Pl.polyadd(pu, ci * Pl.polypow(lin, i))
    pu[0] += 1.0          # the factor 1/2 of Phi(-a) = erfc(a / sqrt 2) / 2
    return pu.astype(np.float32)


def check(pu, X):
    xs = np.linspace(-12, 12, 2400001)
    a = np.minimum(np.abs(xs), X).astype(np.float32)
    q = np.zeros_like(a) + pu[-1]
    for c in pu[-2::-1]:
        q = (q * a + c).astype(np.float32)
    got = np.maximum(xs, 0).astype(np.float32) - a * np.exp2(-q, dtype=np.float32)
    err = np.abs(got - xs * 0.5 * erfc(-xs / np.sqrt(2)))
    return err.max(), xs[err.argmax()]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--deg", type=int, default=8)
    ap.add_argument("--X", type=float, default=5.5)
    a = ap.parse_args()
    pu = fit(a.deg, a.X)
    print("coefficients, constant first:", ", ".join(f"{float(v):.9e}f" for v in pu))
    e, at = check(pu, a.X)
    print(f"max |error| {e:.3e} at x = {at:.4f}")
