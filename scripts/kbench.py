#!/usr/bin/env python3
"""Kernel microbench: the fused temporal kernel on a device-resident cube, all tuning arms
interleaved in one process (A/B rule: N variants x M rounds, report median and min).

    python scripts/kbench.py [--dtype f64] [--ny 215 --nx 1440] [--T 8760] [--rounds 7]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from aggfly_amd import hip, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--ny", type=int, default=215)
    ap.add_argument("--nx", type=int, default=1440)
    ap.add_argument("--T", type=int, default=8760)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--plan", default="c2")
    ap.add_argument("--tunings", default="0,1,2,4,16,32")
    ap.add_argument("--spd", type=int, default=24, help="steps per inner group (24 hourly, 1 daily, 2 tmin/tmax)")
    ap.add_argument("--periods", type=int, default=1, help="outer periods (years)")
    ap.add_argument("--regions", type=int, default=3100)
    ap.add_argument("--skew", default=None, choices=[None, "lognormal"],
                    help="lognormal: region sizes over four decades with one row of >= 1e5 cells (synth.skewed_weights_table)")
    ap.add_argument("--data", default="iid", choices=["iid", "era5"],
                    help="iid: 15 + N(0, 12) per element (every wave meets every branch); era5: seasonal + diurnal cycle + N(0, 3), "
                         "the synthetic field of SURVEY.md 8(d) (neighbouring cells and days resemble each other, as in real data)")
    a = ap.parse_args()
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    C = a.ny * a.nx
    g = torch.Generator(device="cuda").manual_seed(1)
    cube = torch.empty((a.T, a.ny, a.nx), dtype=dt, device="cuda")
    # filled on device in slabs: 15 + N(0, 12)
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
    G1 = len(ib) - 1
    ob = np.round(np.linspace(0, G1, a.periods + 1)).astype(np.int64)
    if a.plan == "c4":      # 13 temperature bins per year straight on daily data (single level)
        edges = np.arange(-20, 50, 5.0)
        cols = [dict(inner="bins", inner_args=(edges[i], edges[i + 1], 0)) for i in range(13)]
        ib, ob = ib[ob], np.arange(a.periods + 1, dtype=np.int64)
    elif a.plan == "c5":    # sine-interpolated degree days from (tmin, tmax) pairs, annual sum
        cols = [dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum")]
    elif a.plan == "c2":
        cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
        cols += [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    elif a.plan == "c1":
        cols = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2)]
    else:
        raise SystemExit("plan must be c1, c2, c4 or c5")
    wdf = synth.weights_table(a.ny, a.nx, a.regions, seed=7, skew=a.skew)
    rl = wdf.groupby("index_right").size()
    print(f"weights table: {len(rl)} regions, {len(wdf)} entries, row lengths min {rl.min()} median {int(rl.median())} max {rl.max()}", flush=True)
    R = int(wdf["index_right"].max()) + 1
    csr = hip.CSR(wdf["index_right"].to_numpy(), wdf["cell_id"].to_numpy(), wdf["weight"].to_numpy(), R, C)
    code = hip.F64 if a.dtype == "f64" else hip.F32
    bytes_alg = a.T * C * cube.element_size()
    plans = {}
    for t in [int(x) for x in a.tunings.split(",")]:
        try:
            plans[t] = hip.FusedPlan(a.T, C, code, ib, ob, cols, tuning=t)
            print(f"tuning {t}: {plans[t].describe()}", flush=True)
        except Exception as e:  # variant not in the menu
            print(f"tuning {t}: unavailable ({e})", flush=True)
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
                    print(f"tuning {t}: max rel diff vs first arm {err:.2e}", flush=True)
                continue
            res[t].append(out["kernel_ms"][0])
            tot[t].append(out["kernel_ms"][1])
    rows = []
    for t in plans:
        med, mn = float(np.median(res[t])), float(np.min(res[t]))
        rows.append({"tuning": t, "variant": plans[t].describe().split()[0], "temporal_ms_med": med, "temporal_ms_min": mn,
                     "GBps_med": bytes_alg / med / 1e6, "GBps_best": bytes_alg / mn / 1e6,
                     "sequence_ms_med": float(np.median(tot[t])),
                     "cell_steps_per_s": a.T * C / (float(np.median(tot[t])) / 1e3)})
        print(json.dumps(rows[-1]), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/kbench_{a.plan}_{a.dtype}_{a.ny}x{a.nx}{'' if a.data == 'iid' else '_' + a.data}{'' if not a.skew else '_' + a.skew}.json", "w") as f:
        json.dump({"args": vars(a), "bytes": bytes_alg, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
