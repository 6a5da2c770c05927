#!/usr/bin/env python3
"""cProfile of `dataset_from_path(device="cuda")` on the BASELINE configs[0] store (GPU decode): what is spent outside the read itself."""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aggfly_amd as af
from aggfly_amd import synth
T, ny, nx = 8760, 104, 236
arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15)
ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                       "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    store = os.path.join(d, "s.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 24, "latitude": ny, "longitude": nx}, compress="blosc")
    fn = lambda: af.dataset_from_path(store, "t2m", lon_is_360=False, device="cuda", preprocess=lambda x: x - 273.15)
    for _ in range(3):
        fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("dataset_from_path ms:", [round(t, 2) for t in sorted(ts)])
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        fn(); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
