#!/usr/bin/env python3
"""Generates tests/golden/sine_dd_fixtures.json: known answers for sine-interpolated degree days at 50 digits.

    python tests/golden/make_sine_fixtures.py          (build container: needs mpmath, which the GPU box need not have)

The reference holds no numeric sine_dd vector (K1 / K3 only compare its two engines with each other at
1e-9), so the closed forms of `_block_sine_dd` (`aggfly/aggregate/nb_kernels.py:202-251`; dask twins
`aggfly/aggregate/temporal.py:313-391`) are evaluated here in 50-digit arithmetic on the EXACT values of the inputs:

    window -> tmin, tmax, tavg = sum / n  (exact rationals of the stored doubles)
    cooling part(thr):  thr <= tmin -> tavg - thr;  tmin < thr < tmax -> ((tavg - thr) a + rng sin(a) / 2) / pi,
                        a = acos((2 thr - tmax - tmin) / rng);  else 0                          kind 0: part(t0) - part(t1)
    heating part(thr):  thr >= tmax -> thr - tavg;  tmin < thr < tmax -> ((thr - tavg)(at + pi / 2) + alpha cos(at)) / pi,
                        alpha = rng / 2, r = (thr - tavg) / alpha, at = atan(r / sqrt(1 - r^2));  else 0   kind 1: part(t1) - part(t0)
                        (|r| > 1 — possible when the window's mean is not its mid-range — is NaN in the reference: sqrt of a negative)

These are data (inputs + expected outputs), not source: the test files hold each implementation to them and state the
bound it achieves.  Cases:
  * windows of 2 rows ((tmin, tmax) pairs: BASELINE configs[4]), 4 rows (6-hourly) and 6 rows (generic group end);
  * thresholds far outside, at the edges, and inside the window; thresholds within 1e-3 ... 1e-9 of the window's range
    from either edge (where acos / sqrt(1 - r^2) lose digits: the cases that decide which side carries an error);
  * `f32_ok` cases have float32-representable windows and run on float32 cubes too.
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))

DDARGS = [[10.0, 30.0, 0.0], [10.0, 30.0, 1.0], [0.0, 5.0, 0.0], [-5.5, 12.25, 1.0], [29.0, 31.0, 0.0],
          [18.3, 18.3000001, 0.0], [-40.0, 60.0, 1.0], [25.0, 8.0, 0.0]]      # the last one reversed (t0 > t1): the forms must not assume an order


def exact(window, t0, t1, kind):
    w = [mp.mpf(float(v)) for v in window]
    tmin, tmax = min(w), max(w)
    tavg = mp.fsum(w) / len(w)
    rng = tmax - tmin

    def cool(thr):
        if thr <= tmin:
            return tavg - thr
        if tmin < thr < tmax:
            a = mp.acos((2 * thr - tmax - tmin) / rng)
            return ((tavg - thr) * a + rng * mp.sin(a) / 2) / mp.pi
        return mp.mpf(0)

    def heat(thr):
        if thr >= tmax:
            return thr - tavg
        if tmin < thr < tmax:
            alpha = rng / 2
            r = (thr - tavg) / alpha
            if abs(r) > 1:
                return mp.nan
            if abs(r) == 1:
                at = mp.pi / 2 * mp.sign(r)
                return ((thr - tavg) * (at + mp.pi / 2)) / mp.pi
            at = mp.atan(r / mp.sqrt(1 - r * r))
            return ((thr - tavg) * (at + mp.pi / 2) + alpha * mp.cos(at)) / mp.pi
        return mp.mpf(0)

    t0, t1 = mp.mpf(float(t0)), mp.mpf(float(t1))
    return cool(t0) - cool(t1) if kind == 0 else heat(t1) - heat(t0)


def main():
    rng = np.random.default_rng(20261005)
    cases = []

    def add(window, row, f32_ok, tag):
        window = [float(v) for v in window]
        t0, t1, kind = DDARGS[row]
        v = exact(window, t0, t1, int(kind))
        if mp.isnan(v):
            # |r| > 1: keep only cases that are safely NaN in double arithmetic too
            w = np.array(window); alpha = (w.max() - w.min()) / 2
            rs = [abs((t - w.mean()) / alpha) for t in (t0, t1) if w.min() < t < w.max()]
            if not rs or min(abs(r - 1) for r in rs) < 1e-6:
                return
            cases.append({"window": window, "row": row, "f32_ok": bool(f32_ok), "tag": tag, "value": None})
            return
        if len(window) > 2 and int(kind) == 1:
            w = np.array(window); alpha = (w.max() - w.min()) / 2
            rs = [abs((t - w.mean()) / alpha) for t in (t0, t1) if w.min() < t < w.max()]
            if rs and min(abs(r - 1) for r in rs) < 1e-6:
                return                                           # |r| within 1e-6 of 1: NaN or not depends on the last bit of tavg
        cases.append({"window": window, "row": row, "f32_ok": bool(f32_ok), "tag": tag, "value": mp.nstr(v, 25), "value_f64": float(v)})

    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    # 1. random windows around the thresholds (float32-representable): the bulk
    for L, n in ((2, 700), (4, 250), (6, 250)):
        for _ in range(n):
            row = int(rng.integers(0, len(DDARGS)))
            t0, t1, _k = DDARGS[row]
            centre = rng.choice([t0, t1, 0.5 * (t0 + t1)]) + rng.normal(0, 6)
            w = f32(centre + rng.normal(0, rng.choice([0.3, 3.0, 12.0]), L))
            if w.max() == w.min():
                continue
            add(w, row, True, f"random L={L}")
    # 2. thresholds AT the window's edges and at its mean (the branch boundaries of the closed forms)
    for row, (t0, t1, _k) in enumerate(DDARGS):
        for thr in (t0, t1):
            if float(np.float32(thr)) != thr:
                continue
            for span in (0.5, 7.0, 33.0):
                s = float(np.float32(span))
                add([thr, thr + s], row, True, "threshold == tmin")
                add([thr - s, thr], row, True, "threshold == tmax")
                add([thr - s, thr + s], row, True, "threshold == mid-range")
                add([thr - s, thr - s / 2, thr + s / 2, thr + s], row, True, "threshold == mean, L=4")
    # 3. thresholds within delta * rng of either edge, delta = 1e-3 ... 1e-9 (float64 windows; 2, 4 and 6 rows)
    for row, (t0, t1, _k) in enumerate(DDARGS):
        for thr in (t0, t1):
            for e in range(3, 10):
                for _ in range(4):
                    delta = 10.0 ** (-e) * rng.uniform(1, 9.99)
                    span = rng.choice([0.7, 5.0, 21.0]) * rng.uniform(0.8, 1.25)
                    for side in (0, 1):
                        tmin = thr - delta * span if side == 0 else thr - (1 - delta) * span
                        tmax = tmin + span
                        L = int(rng.choice([2, 2, 4, 6]))
                        inner = rng.uniform(tmin, tmax, L - 2)
                        w = np.concatenate([[tmin], inner, [tmax]])
                        add(w[rng.permutation(L)] if L > 2 else w, row, False, f"threshold {delta:.1e} * rng inside {'tmin' if side == 0 else 'tmax'}")
    out = {"about": "sine_dd known answers at 50 digits; generated by tests/golden/make_sine_fixtures.py (mpmath), closed forms of "
                    "aggfly/aggregate/nb_kernels.py:202-251 on the exact values of the inputs; value = null: the reference gives NaN (|r| > 1)",
           "ddargs": DDARGS, "cases": cases}
    path = os.path.join(HERE, "sine_dd_fixtures.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    tags = {}
    for c in cases:
        k = c["tag"].split(" * ")[-1] if c["tag"].startswith("threshold ") and "*" in c["tag"] else c["tag"]
        tags[k] = tags.get(k, 0) + 1
    print(f"{len(cases)} cases -> {path} ({os.path.getsize(path) >> 10} KiB)")
    for k, v in sorted(tags.items()):
        print(f"  {v:5d}  {k}")


if __name__ == "__main__":
    main()
