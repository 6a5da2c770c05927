#!/usr/bin/env python3
"""Host emulation (numpy float64) of round 3's table-driven sine arc (afhip_sine.h: sine_theta / sine_arc) against the
reference's acos / sin / atan / cos closed forms (oracle.ref_temporal): checks the algebra and the error budget before a
GPU run.  rcp / rsq seeds are emulated with a relative error of 2^-23.

  theta = acos(a), a in [0, 1]:  g = sqrt((1 - a)(1 + a));  u = min(a, g) <= 0.7072, w = max(a, g);  k = round(u * S);
  phi_k = asin(k / S):  asin(u) = phi_k + asin(delta),  delta = u cos(phi_k) - w sin(phi_k)   (|delta| <= ~1 / (2 S cos))
  a <= g:  theta = pi/2 - asin(u) = (pi/2 - phi_k) + asin(w sin(phi_k) - u cos(phi_k))
  a >  g:  theta = asin(u)        = phi_k + asin(u cos(phi_k) - w sin(phi_k))
so one table row (C, S, THETA) per (half, k) gives theta = THETA + asin(u C + w S).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_temporal as rt  # noqa: E402

rng = np.random.default_rng(5)
SCALE = int(os.environ.get("SCALE", "256"))
NCOEF = int(os.environ.get("NCOEF", "2"))        # asin(delta) = delta + delta^3 (1/6 + 3/40 delta^2) or just 1/6
SQRT_FIX = int(os.environ.get("SQRT_FIX", "0"))  # 1: the residual correction after the Goldschmidt step
NK = int(np.floor(0.70710678118654757 * SCALE + 0.5)) + 2


def table():
    k = np.arange(NK)
    s = k / SCALE
    c = np.sqrt((1 - s) * (1 + s))
    phi = np.arcsin(s)
    big = np.stack([c, -s, phi], axis=1)                 # a > g: theta = phi + asin(u c - w s)
    small = np.stack([-c, s, np.pi / 2 - phi], axis=1)   # a <= g: theta = (pi/2 - phi) + asin(w s - u c)
    return small, big


T_SMALL, T_BIG = table()


def noisy(x):
    return x * (1 + rng.uniform(-1, 1, x.shape) * 2.0 ** -23)


def theta_g(a):
    """acos(a) and sqrt(1 - a^2) for a in [0, 1)."""
    q = np.maximum((1 - a) * (1 + a), 2.2250738585072014e-308)
    y = noisy(1 / np.sqrt(q))
    g = q * y
    h = 0.5 * y
    r = 0.5 - h * g
    g = g + g * r
    if SQRT_FIX:
        h = h + h * r
        g = g + (q - g * g) * h
    u, w = np.minimum(a, g), np.maximum(a, g)
    small = a <= g
    k = np.floor(u * SCALE + 0.5).astype(np.int64)
    k = np.clip(k, 0, NK - 1)
    row = np.where(small[:, None], T_SMALL[k], T_BIG[k])
    delta = u * row[:, 0] + w * row[:, 1]
    t = delta * delta
    p = (1 / 6 + 0.075 * t) if NCOEF == 2 else np.full_like(t, 1 / 6)
    th = row[:, 2] + (delta + (delta * t) * p)
    return th, g, np.abs(delta).max()


def parts_pair(thr, tmin, tmax):
    """(cooling part, heating part) of one threshold on (tmin, tmax) pairs: max(+-d, 0) + [inside] alpha F(a)."""
    tavg = (tmin + tmax) * 0.5
    alpha = (tmax - tmin) * 0.5
    with np.errstate(all="ignore"):
        y0 = noisy(1 / alpha)
        y = y0 + y0 * (1 - alpha * y0)           # one Newton step
        d = thr - tavg
        a = np.abs(d) * y
        th, g, dmax = theta_g(a)
        arc = (alpha * 0.31830988618379067154) * (g - a * th)
    inside = (thr < tmax) & (tmin < thr)
    cool = np.maximum(-d, 0.0) + np.where(inside, arc, 0.0)
    heat = np.maximum(d, 0.0) + np.where(inside, arc, 0.0)
    return cool, heat, dmax


def main():
    n = 4_000_000
    w = rng.normal(18, 9, (2, n))
    if os.environ.get("F32"):
        w = w.astype(np.float32).astype(np.float64)
    tmin, tmax = w.min(0), w.max(0)
    tavg = (tmin + tmax) * 0.5
    worst = {}
    for thr in (10.0, 30.0, 20.5, 5.0, 18.0, 0.0):
        cool, heat, dmax = parts_pair(thr, tmin, tmax)
        rc = rt._sine_part_cooling(thr, tmin, tmax, tavg)
        rh = rt._sine_part_heating(thr, tmin, tmax, tavg)
        for nm, got, want in (("cool", cool, rc), ("heat", heat, rh)):
            assert np.array_equal(np.isnan(got), np.isnan(want)), (nm, thr)
            err = np.abs(got - want)
            big = np.abs(want) > 1e-6
            rel = (err[big] / np.abs(want[big])).max()
            worst[(nm, thr)] = (err.max(), rel)
            print(f"{nm} thr={thr:5.1f}: max abs err {err.max():.3e}  max rel err (|want| > 1e-6) {rel:.3e}  |delta| max {dmax:.5f}")
    print("SCALE", SCALE, "rows per half", NK, "LDS bytes", 2 * NK * 32, "NCOEF", NCOEF, "SQRT_FIX", SQRT_FIX,
          "| worst abs", max(v[0] for v in worst.values()), "worst rel", max(v[1] for v in worst.values()))


if __name__ == "__main__":
    main()
