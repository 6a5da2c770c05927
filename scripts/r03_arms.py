#!/usr/bin/env python3
"""A/B of library knobs on one plan in ONE process: every arm is a FusedPlan created under its own environment
(the library reads its experiment knobs at plan creation) and / or tuning code; rounds are interleaved, medians reported.

    python scripts/r03_arms.py --plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600 \
        --arms "base" "AFHIP_WGS_PER_CU=8" "AFHIP_WGS_PER_CU=8,AFHIP_FORCE_WG=64" "tuning=208"
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from aggfly_amd import hip, synth  # noqa: E402


def columns(plan):
    if plan == "c4":
        edges = np.arange(-20, 50, 5.0)
        return [dict(inner="bins", inner_args=(edges[i], edges[i + 1], 0)) for i in range(13)], True
    if plan == "c5":
        return [dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum")], False
    if plan == "c2":
        return [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")] + \
            [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)], False
    if plan == "c1":
        return [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2)], False
    if plan == "dd":        # degree days alone: a threshold slot and no statistic
        return [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")], False
    if plan == "dd13":      # thirteen degree-day columns (multi_dd with thirteen threshold pairs)
        return [dict(inner="dd", inner_args=(float(t), float(t) + 7, 0), outer="sum") for t in range(-10, 29, 3)], False
    if plan == "mean":      # daily mean of short groups -> annual sum (tmin/tmax pairs, 6-hourly data)
        return [dict(inner="mean", outer="sum")], False
    if plan == "ref":       # the reference's published benchmark shape: mean@date -> power[1..4] -> sum@month (use --periods 12)
        return [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)], False
    if plan == "meanpoly":  # the polynomial of the daily mean (of tmin / tmax pairs, of 6-hourly steps ...) -> annual sum
        return [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)], False
    raise SystemExit("plan must be c1, c2, c4, c5, mean or meanpoly")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--ny", type=int, default=215)
    ap.add_argument("--nx", type=int, default=1440)
    ap.add_argument("--T", type=int, default=8760)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--plan", default="c2")
    ap.add_argument("--spd", type=int, default=24)
    ap.add_argument("--periods", type=int, default=1)
    ap.add_argument("--gaps", type=float, default=0.0, help="share of the inner groups that miss one step (a series with gaps: mixed group lengths)")
    ap.add_argument("--regions", type=int, default=3100)
    ap.add_argument("--data", default="era5", choices=["iid", "era5"])
    ap.add_argument("--arms", nargs="+", default=["base"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--sustained", type=int, default=0, help="also time N launches of every arm back to back WITHOUT a sync between them "
                                                             "(HIP-event pairs around the temporal kernel, as bench.py does)")
    a = ap.parse_args()
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    C = a.ny * a.nx
    g = torch.Generator(device="cuda").manual_seed(1)
    cube = torch.empty((a.T, a.ny, a.nx), dtype=dt, device="cuda")
    lat = torch.linspace(0.6, 1.4, a.ny, device="cuda", dtype=torch.float32)[None, :, None]
    for k0 in range(0, a.T, 512):
        k1 = min(a.T, k0 + 512)
        noise = torch.randn((k1 - k0, a.ny, a.nx), generator=g, device="cuda", dtype=torch.float32)
        if a.data == "iid":
            cube[k0:k1] = (15 + 12 * noise).to(dt)
        else:
            k = torch.arange(k0, k1, device="cuda", dtype=torch.float32)
            base = 15.0 + 12.0 * torch.sin(2 * np.pi * torch.floor(k / a.spd) / 365.0) + 6.0 * torch.sin(2 * np.pi * (k % a.spd) / a.spd - np.pi / 2)
            cube[k0:k1] = (base[:, None, None] * lat + 3.0 * noise).to(dt)
    ib = synth.hourly_bounds(a.T, a.spd)
    if a.gaps > 0:
        lens = np.diff(ib)
        hit = np.random.default_rng(3).random(len(lens)) < a.gaps
        lens = np.where(hit & (lens > 1), lens - 1, lens)
        ib = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        a.T = int(ib[-1])
        cube = cube[:a.T]
    G1 = len(ib) - 1
    ob = np.round(np.linspace(0, G1, a.periods + 1)).astype(np.int64)
    cols, single = columns(a.plan)
    if single:
        ib, ob = ib[ob], np.arange(a.periods + 1, dtype=np.int64)
    wdf = synth.weights_table(a.ny, a.nx, a.regions, seed=7)
    R = int(wdf["index_right"].max()) + 1
    csr = hip.CSR(wdf["index_right"].to_numpy(), wdf["cell_id"].to_numpy(), wdf["weight"].to_numpy(), R, C)
    code = hip.F64 if a.dtype == "f64" else hip.F32
    bytes_alg = a.T * C * cube.element_size()
    plans = {}
    for arm in a.arms:
        env, tuning = {}, 0
        for kv in ([] if arm == "base" else arm.split(",")):
            k, v = kv.split("=")
            if k == "tuning":
                tuning = int(v)
            else:
                env[k] = v
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            plans[arm] = hip.FusedPlan(a.T, C, code, ib, ob, cols, tuning=tuning)
            print(f"{arm}: {plans[arm].describe()}", flush=True)
        except Exception as e:
            print(f"{arm}: unavailable ({e})", flush=True)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    res = {t: [] for t in plans}
    tot = {t: [] for t in plans}
    ref = None
    for r in range(a.rounds + 1):
        for t, p in plans.items():
            out = p.run(cube, csr, timed=True)
            if r == 0:
                v = out["res"].cpu().numpy()
                if ref is None:
                    ref = v
                else:
                    err = np.nanmax(np.abs(v - ref) / np.maximum(np.abs(ref), 1e-300))
                    print(f"{t}: max rel diff vs first arm {err:.2e}", flush=True)
                continue
            res[t].append(out["kernel_ms"][0])
            tot[t].append(out["kernel_ms"][1])
    sustained = {}
    if a.sustained:
        for t, p in plans.items():
            outb = p.run(cube, csr)
            for _ in range(10):
                p.run(cube, csr, out=outb)
            torch.cuda.synchronize()
            p.profile_begin(a.sustained)
            for _ in range(a.sustained):
                p.run(cube, csr, out=outb)
            torch.cuda.synchronize()
            sustained[t] = float(np.mean(p.profile_end()))
    rows = []
    for t in plans:
        med, mn = float(np.median(res[t])), float(np.min(res[t]))
        rows.append({"arm": t, "variant": plans[t].describe().split()[0], "temporal_ms_med": round(med, 4), "temporal_ms_min": round(mn, 4),
                     "GBps_med": round(bytes_alg / med / 1e6, 1), "frac_of_8TBps": round(bytes_alg / med / 1e6 / 8000, 3),
                     "sequence_ms_med": round(float(np.median(tot[t])), 4), "describe": plans[t].describe()})
        if a.sustained:
            rows[-1]["sustained_ms_mean"] = round(sustained[t], 4)
            rows[-1]["sustained_frac_of_8TBps"] = round(bytes_alg / sustained[t] / 1e6 / 8000, 3)
        print(json.dumps({k: v for k, v in rows[-1].items() if k != "describe"}), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump({"args": vars(a), "bytes": bytes_alg, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
