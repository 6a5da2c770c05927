"""Largest deviation of the device sine_dd closed forms from the CPU restatement of the reference
(oracle/c, libm acos/sin/atan/cos) on random windows: hourly (24-step) and tmin/tmax (2-step)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aggfly_amd import hip          # noqa: E402
from oracle import cport            # noqa: E402

out = {}
rng = np.random.default_rng(11)
for name, T, step in (("hourly", 24 * 400, 24), ("minmax_pairs", 2 * 4000, 2)):
    ny, nx = 8, 64
    cube = rng.normal(18, 9, (T, ny, nx))
    bounds = np.arange(0, T + 1, step, dtype=np.int64)
    dda = [[10, 30, 0], [5, 18, 1], [20, 21, 0], [-5, 40, 1]]
    want = cport.block_sine_dd(cube, bounds, dda)
    got = hip.group_sine_dd(torch.from_numpy(cube).cuda(), bounds, dda).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = np.isfinite(want)
    err = np.abs(got[m] - want[m])
    rel = err / np.maximum(np.abs(want[m]), 1e-300)
    scale = np.abs(want[m]).max()
    out[name] = {"values": int(m.sum()), "max_abs_err": float(err.max()), "max_rel_err_where_abs_gt_1e-6": float(rel[np.abs(want[m]) > 1e-6].max()),
                 "max_abs_err_over_scale": float(err.max() / scale)}
print(json.dumps(out, indent=1))
