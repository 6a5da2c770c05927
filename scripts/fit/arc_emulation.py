#!/usr/bin/env python3
"""Host emulation (numpy float64) of the kernel's sine_arc closed form (afhip_sine.h) against the reference's
acos / sin / atan / cos form (oracle.ref_temporal): checks the algebra and the error budget before a GPU run.
The rsq seed is emulated with a relative error of 2^-23."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_temporal as rt

C = [0.3379097149030259, -0.9032914986427146, 1.1675177613361616, -0.890631195996176, 0.4674407127226085, -0.15540291391580394,
     0.05097480101568288, 0.0042099716965424624, 0.014879304155047953, 0.0172753040904813, 0.022376412991636434,
     0.030381804553144803, 0.0446428595401402, 0.07499999998385828, 0.16666666666668462]
rng = np.random.default_rng(5)


def arc(d, x, alpha):
    a = np.abs(x)
    q = (1 - a) * (1 + a)
    with np.errstate(invalid="ignore", divide="ignore"):
        y = 1 / np.sqrt(q + 2.2250738585072014e-308) * (1 + rng.uniform(-1, 1, q.shape) * 2.0 ** -23)
    g = q * y; h = 0.5 * y
    r = 0.5 - h * g
    g = g + g * r; h = h + h * r
    e = q - g * g
    g = g + e * h
    small = a <= 0.70710678118654752
    u = np.where(small, a, g)
    t = u * u
    p = np.full_like(t, C[0])
    for c in C[1:]:
        p = p * t + c
    as_ = u + (u * t) * p
    v = np.where(small, as_, np.pi / 2 - as_)
    ac = np.pi / 2 - np.copysign(v, x)
    return d * ac + alpha * g


def cool(thr, tmin, tmax, tavg):
    alpha = (tmax - tmin) * 0.5
    with np.errstate(all="ignore"):
        inv = 1 / (tmax - tmin)
        z = (2 * thr - tmax - tmin) * inv
        inwin = (thr < tmax) & (tmin < thr)
        return np.where(thr <= tmin, tavg - thr, np.where(inwin, arc(tavg - thr, z, alpha) * (1 / np.pi), 0.0))


def heat(thr, tmin, tmax, tavg):
    alpha = (tmax - tmin) * 0.5
    with np.errstate(all="ignore"):
        inv_a = 2 * (1 / (tmax - tmin))
        d = thr - tavg
        inwin = (thr < tmax) & (tmin < thr)
        return np.where(thr >= tmax, thr - tavg, np.where(inwin, arc(d, -(d * inv_a), alpha) * (1 / np.pi), 0.0))


for name, nstep in (("pairs", 2), ("hourly", 24)):
    w = rng.normal(18, 9, (nstep, 2_000_000))
    tmin, tmax, tavg = w.min(0), w.max(0), w.mean(0)
    for thr in (10.0, 30.0, 20.5, 5.0):
        with np.errstate(all="ignore"):
            want_c = rt._sine_part_cooling(thr, tmin, tmax, tavg)
            want_h = rt._sine_part_heating(thr, tmin, tmax, tavg)
        for kind, got, want in (("cool", cool(thr, tmin, tmax, tavg), want_c), ("heat", heat(thr, tmin, tmax, tavg), want_h)):
            assert np.array_equal(np.isnan(got), np.isnan(want)), (name, thr, kind, np.isnan(got).sum(), np.isnan(want).sum())
            m = np.isfinite(want)
            err = np.abs(got[m] - want[m])
            big = np.abs(want[m]) > 1e-6
            print(f"{name:7s} thr {thr:5.1f} {kind}: max abs {err.max():.2e}  max rel (|x| > 1e-6) {(err[big] / np.abs(want[m][big])).max():.2e}  NaN {int((~m).sum())}")
