#!/usr/bin/env python3
"""Where the decode-in-HBM route passes the host-thread route on stores whose chunks hold whole time steps (io.GPU_DECODE_AUTO_BYTES_WHOLE_ROWS): store -> HBM, best / median
of 7 reads, both routes forced, request sizes from 0.1 to 1.2 GB (104 x 236 f32, 24-step chunks — LAYOUT=tiled: 504 x 52 x 118 —, Blosc-LZ4 + shuffle; FIELD=noisy|smooth|smooth_noise)."""
import os, sys, tempfile, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aggfly_amd as af
from aggfly_amd import synth

ny, nx = 104, 236
for hours in [int(h) for h in os.environ.get("HOURS_LIST", "1008,1512,2016,2520,3504,5016,8760,12000").split(",")]:
    T = hours
    arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15)
    if os.environ.get("FIELD", "noisy") != "noisy":
        k = np.arange(T)[:, None, None]; y = np.arange(ny)[None, :, None]; x = np.arange(nx)[None, None, :]
        smooth = 285 + 12 * np.sin(2 * np.pi * k / 8760.0) + 5 * np.sin(2 * np.pi * (k % 24) / 24) + 8 * np.sin(y / 17.0) * np.cos(x / 23.0)
        if os.environ["FIELD"] == "smooth_noise":
            smooth = smooth + np.random.default_rng(1).normal(0, 0.3, smooth.shape)
        arr = (np.round(smooth * 100) / 100).astype(np.float32)
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                           "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        store = os.path.join(d, "s.zarr")
        chunks = {"time": 24, "latitude": ny, "longitude": nx} if os.environ.get("LAYOUT", "rows") == "rows" else {"time": 504, "latitude": 52, "longitude": 118}
        af.dataset_to_zarr(ds, store, var="t2m", chunks=chunks, compress="blosc")
        fn = lambda: af.dataset_from_path(store, "t2m", lon_is_360=False, device="cuda")
        row = []
        for mode in ("1", "0"):
            os.environ["AGGFLY_HIP_GPU_DECODE"] = mode
            os.environ["AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MIN_MB"] = "64"
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            row.append((min(ts), sorted(ts)[3]))
        print(f"{arr.nbytes / 1e6:7.0f} MB decoded   in HBM {row[0][0]:6.2f} / {row[0][1]:6.2f} ms   host threads {row[1][0]:6.2f} / {row[1][1]:6.2f} ms", flush=True)
