#!/usr/bin/env python3
"""Why does ONE plan replayed back to back run slower than the same plan interleaved with another one (f32 K = 5: 1.90 vs 1.77 ms)?
Per-launch times of the configs[1] plan on f32 storage: replayed alone, with a 2 GB device memset between launches, with a host sleep
between launches, and alternating with the d8 arm."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aggfly_amd import hip, synth

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
T, ny, nx = 8760, 215, 1440
C = ny * nx
dt = torch.float32 if dtype == "f32" else torch.float64
g = torch.Generator(device="cuda").manual_seed(1)
cube = torch.empty((T, ny, nx), dtype=dt, device="cuda")
for k0 in range(0, T, 512):
    k1 = min(T, k0 + 512)
    cube[k0:k1] = (15 + 12 * torch.randn((k1 - k0, ny, nx), generator=g, device="cuda", dtype=torch.float32)).to(dt)
ib = synth.hourly_bounds(T)
ob = np.array([0, len(ib) - 1], dtype=np.int64)
cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")] + [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
wdf = synth.weights_table(ny, nx, 3100, seed=7)
R = int(wdf["index_right"].max()) + 1
csr = hip.CSR(wdf["index_right"].to_numpy(), wdf["cell_id"].to_numpy(), wdf["weight"].to_numpy(), R, C)
code = hip.F32 if dtype == "f32" else hip.F64
a = hip.FusedPlan(T, C, code, ib, ob, cols)
b = hip.FusedPlan(T, C, code, ib, ob, cols, tuning=208 if dtype == "f32" else 108)
print(a.describe().split()[0], "|", b.describe().split()[0])
junk = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")


def series(label, between):
    ts = []
    a.run(cube, csr, timed=True)
    for _ in range(12):
        between()
        ts.append(a.run(cube, csr, timed=True)["kernel_ms"][0])
    print(f"{label:34s} median {np.median(ts):.3f}  min {min(ts):.3f}  max {max(ts):.3f} | " + " ".join(f"{t:.3f}" for t in ts), flush=True)


series("replayed alone", lambda: None)
series("2 GB memset between", lambda: (junk.zero_(), torch.cuda.synchronize()))
series("50 ms host sleep between", lambda: time.sleep(0.05))
series("5 ms host sleep between", lambda: time.sleep(0.005))
series("other arm between", lambda: b.run(cube, csr, timed=True))
series("untimed run of itself between", lambda: (a.run(cube, csr), torch.cuda.synchronize()))
series("replayed alone (again)", lambda: None)
