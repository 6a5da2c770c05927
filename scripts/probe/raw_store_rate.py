#!/usr/bin/env python3
"""Uncompressed Zarr store (RAM) -> HBM, decoded GB/s, best of 4: the raw chunk files go through the same piecewise reader as
the compressed ones of the decode-in-HBM route (AFCODEC_NT_COPY=0|1 switches its copy)."""
import os, sys, tempfile, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aggfly_amd as af
from aggfly_amd import synth
T, ny, nx = 8760 * int(os.environ.get("YEARS", "1")), 104, 236
arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1)
ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                       "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    for lname, chunks in (("tiled 744x52x118", {"time": 744, "latitude": 52, "longitude": 118}), ("rows 24 x grid", {"time": 24, "latitude": ny, "longitude": nx})):
        store = os.path.join(d, "s.zarr")
        af.dataset_to_zarr(ds, store, var="t2m", chunks=chunks, compress=False)
        fn = lambda: af.dataset_from_path(store, "t2m", lon_is_360=False, device="cuda")
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); got = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert np.array_equal(got.cube()[:3].cpu().numpy(), arr[:3])
        print(f"raw store, {lname}: {arr.nbytes / 1e9 / best:.1f} GB/s (AFCODEC_NT_COPY={os.environ.get('AFCODEC_NT_COPY', 'default')})", flush=True)
        import shutil; shutil.rmtree(store)
