#!/usr/bin/env python3
"""Wall time of `af.aggregate_dataset` for a DAILY panel (groupby date as the last step) on a resident cube: hourly f32, 215 x 1440, 3,100 regions,
dd[10,30]@date + mean@date -> power[1..4]: 365 x 3,100 rows x 5 columns.  Kernel time against the whole call (spec lowering, plan cache, frame assembly, merge)."""
import json, os, sys, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aggfly_amd as af
from aggfly_amd import synth

T, ny, nx, R = 8760, 215, 1440, 3100
g = torch.Generator(device="cuda").manual_seed(3)
cube = 15 + 8 * torch.randn((T, ny, nx), generator=g, device="cuda", dtype=torch.float32)
ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                       "latitude": 25 + 0.25 * np.arange(ny), "longitude": 235 + 0.25 * np.arange(nx)}), lon_is_360=True)
tab = synth.weights_table(ny, nx, R, seed=7)
gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i:05d}" for i in range(int(tab.index_right.max()) + 1)]}))
w = af.weights_from_objects(ds, gr, table=tab)
spec = dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]})],
            tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 5)})])
out = {}
for name, sp in (("daily panel (P=365, K=5)", spec),
                 ("annual panel (P=1, K=5)", {k: v + [("aggregate", {"calc": "sum", "groupby": "year"})] for k, v in spec.items()})):
    af.aggregate_dataset(dataset=ds, weights=w, **sp)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        df = af.aggregate_dataset(dataset=ds, weights=w, **sp)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    out[name] = {"rows": len(df), "columns": len(df.columns), "aggregate_dataset_ms": best * 1e3, "cell_steps_per_s": T * ny * nx / best}
print(json.dumps(out, indent=1))
if os.environ.get("API_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        af.aggregate_dataset(dataset=ds, weights=w, **spec)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
