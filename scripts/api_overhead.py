#!/usr/bin/env python3
"""Wall time of the public aggregate_dataset() call on a device-resident configs[1] year:
how much host-side Python sits on top of the 3.4 ms kernel sequence."""
import cProfile, io, json, os, pstats, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aggfly_amd as af
import bench
from aggfly_amd import synth

T, ny, nx = 8760, 215, 1440
cube = bench.make_cube(torch, T, ny, nx, torch.float64, 1)
time_idx = pd.date_range("2001-01-01", periods=T, freq="h")
da = af.DataArray(cube, ["time", "latitude", "longitude"], {"time": time_idx, "latitude": 24 + 0.25 * np.arange(ny), "longitude": 0.25 * np.arange(nx)})
ds = af.Dataset(da, lon_is_360=True)
tab = synth.weights_table(ny, nx, 3100, seed=7)
gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
w = af.weights_from_objects(ds, gr, table=tab)
spec = dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "year"})],
            tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 5)}),
                  ("aggregate", {"calc": "sum", "groupby": "year"})])
af.aggregate_dataset(dataset=ds, weights=w, **spec)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); df = af.aggregate_dataset(dataset=ds, weights=w, **spec); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({"api_call_ms_median": float(np.median(ts)) * 1e3, "api_call_ms_min": float(np.min(ts)) * 1e3, "rows": len(df)}))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    af.aggregate_dataset(dataset=ds, weights=w, **spec)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3500])
