#!/usr/bin/env python3
"""cProfile of a SMALL `dataset_from_path(device="cuda")` request (a 25 MB window of whole-time-step chunks; host-thread decode): what a window read costs beside its bytes."""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aggfly_amd as af
from aggfly_amd import synth
T, ny, nx = 2016, 104, 236
arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15)
ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                       "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    store = os.path.join(d, "s.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 24, "latitude": ny, "longitude": nx}, compress="blosc")
    fn = lambda: af.dataset_from_path(store, "t2m", lon_is_360=False, device="cuda", time_sel=slice("2001-01-10", "2001-01-19"))
    for _ in range(3):
        fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); got = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("window bytes", got.cube().numel() * 4 / 1e6, "MB; ms:", [round(t, 2) for t in sorted(ts)])
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        fn(); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
