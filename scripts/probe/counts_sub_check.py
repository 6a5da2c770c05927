#!/usr/bin/env python3
"""Which of the packed-count gathers is right on a 40,000-region table: one lane per pair (default), a group of lanes per pair (AFHIP_COUNTS_SPMM_SUB), table order (exact_order)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aggfly_amd import hip, synth
P = int(os.environ.get("PERIODS", 2))
T, ny, nx, R0 = 168 * P, int(os.environ.get("NY", 1801)), int(os.environ.get("NX", 360)), int(os.environ.get("REGIONS", 40000))
g = torch.Generator(device="cuda").manual_seed(1)
cube = (15 + 12 * torch.randn((T, ny, nx), generator=g, device="cuda", dtype=torch.float32))
tab = synth.weights_table(ny, nx, R0, seed=7)
R = int(tab["index_right"].max()) + 1
csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, ny * nx)
edges = np.arange(-20, 50, 5.0)
cols = [dict(inner="bins", inner_args=(edges[i], edges[i + 1], 0)) for i in range(13)]
ib = np.arange(0, T + 1, 168, dtype=np.int64); ob = np.arange(P + 1, dtype=np.int64)
res = {}
for name, env, kw in (("exact", {}, dict(exact_order=True)), ("one lane", {}, {}), ("sub 8", {"AFHIP_COUNTS_SPMM_SUB": "8"}, {}), ("no counts gather", {"AFHIP_NO_COUNTS_SPMM": "1"}, {})):
    os.environ.update(env)
    plan = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, cols, **kw)
    for k in env: del os.environ[k]
    out = plan.run(cube, csr)
    res[name] = {k: out[k].cpu().numpy() for k in ("num", "den", "res")}
    print(name, plan.describe()[:150])
for name in ("one lane", "sub 8", "no counts gather"):
    for key in ("num", "den", "res"):
        a, b = res[name][key], res["exact"][key]
        d = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        bad = np.argwhere(np.nan_to_num(d) > 1e-9)
        print(f"{name:18s} {key}: max rel diff vs exact {np.nanmax(d):.2e}; entries off by more than 1e-9: {len(bad)}", bad[:3].tolist())
        if len(bad) or np.nanmax(d) > 1e-13:
            i = tuple(np.unravel_index(np.nanargmax(d), d.shape))
            print("     worst:", i, "exact", b[i], "got", a[i], "den", res["exact"]["den"][i[-2:]] if key != "den" else "")
print("segments:", csr.describe() if hasattr(csr, "describe") else "")
# the worst entry of `sub 8`, term by term
a, b = res["sub 8"]["num"], res["exact"]["num"]
d = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
k, r, p = (int(x) for x in np.unravel_index(np.nanargmax(d), d.shape))
plan = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, cols, exact_order=True)
cells = plan.run(cube, csr, want_cells=True)["cells"]
rows = tab["index_right"].to_numpy(); sel = np.flatnonzero(rows == r)
cid, wt = tab["cell_id"].to_numpy()[sel], tab["weight"].to_numpy()[sel]
cnt = cells[k, p, torch.from_numpy(cid).cuda()].cpu().numpy()
print("region", r, "period", p, "column", k, "entries", len(sel))
for c, w_, n_ in zip(cid, wt, cnt):
    print(f"   cell {c}  weight {w_!r}  count {n_}")
print("   table-order sum", float(np.sum(wt * cnt)), " exact kernel", b[k, r, p], " sub 8", a[k, r, p], " one lane", res["one lane"]["num"][k, r, p])
