#!/usr/bin/env python3
"""Fit p(t) in asin(u) = u + u^3 p(u^2) on 0 <= u <= umax (t = u^2) by Chebyshev interpolation in mpmath, then report the
max error of the double-precision Horner evaluation (what the kernel runs).  Usage: asin_fit.py [umax] [degree...]"""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 50
umax = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7072
degs = [int(x) for x in sys.argv[2:]] or list(range(9, 20))
tmax = mp.mpf(umax) ** 2


def f(t):
    if t == 0:
        return mp.mpf(1) / 6
    u = mp.sqrt(t)
    return (mp.asin(u) - u) / (u * t)


def fit(deg):
    # interpolate at Chebyshev nodes of [0, tmax], convert to monomial coefficients (mp precision)
    n = deg + 1
    nodes = [tmax / 2 * (1 + mp.cos(mp.pi * (2 * k + 1) / (2 * n))) for k in range(n)]
    A = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(n):
            A[i, j] = x ** j
        b[i] = f(x)
    c = mp.lu_solve(A, b)
    return [c[j] for j in range(n)]


us = np.concatenate([np.linspace(0, umax, 200001), np.random.default_rng(1).uniform(0, umax, 300000)])
ref = np.array([float(mp.asin(mp.mpf(float(u)))) for u in us[::25]])
for deg in degs:
    c = [float(x) for x in fit(deg)]
    t = us * us
    p = np.full_like(t, c[-1])
    for k in range(deg - 1, -1, -1):
        p = p * t + c[k]            # numpy has no fma; the kernel's fma is at least as accurate
    val = us + (us * t) * p
    err = np.abs(val[::25] - ref)
    rel = err / np.maximum(ref, 1e-300)
    print(f"deg {deg}: max abs err {err.max():.3e}  max rel {rel[1:].max():.3e}")
    if len(degs) == 1:
        for k in range(deg, -1, -1):
            print(f"    {c[k]!r},")
