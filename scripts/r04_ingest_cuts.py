#!/usr/bin/env python3
"""Store -> HBM with the chunk decode on the GPU: how the request is cut into batches (AGGFLY_HIP_GPU_DECODE_CUTS: batch ends as
fractions of the request) against the default (a quarter-size first and last batch, full ones between).  BASELINE configs[0] store
(8760 x 104 x 236 f32, 365 chunks of 24 steps, Blosc-LZ4 + shuffle, in /dev/shm); best of 7 reads per setting, ms."""
import json, os, sys, tempfile, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aggfly_amd as af
from aggfly_amd import synth

T, ny, nx = int(os.environ.get("HOURS", 8760)), 104, 236
arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15)
if os.environ.get("FIELD", "noisy") != "noisy":        # smooth fields quantised like reanalysis output (scripts/r02_gpu_decode_ratio.py): Blosc ratio 2 - 6
    k = np.arange(T)[:, None, None]; y = np.arange(ny)[None, :, None]; x = np.arange(nx)[None, None, :]
    smooth = 285 + 12 * np.sin(2 * np.pi * k / 8760.0) + 5 * np.sin(2 * np.pi * (k % 24) / 24) + 8 * np.sin(y / 17.0) * np.cos(x / 23.0)
    if os.environ["FIELD"] == "smooth_noise":
        smooth = smooth + np.random.default_rng(1).normal(0, 0.3, smooth.shape)
    arr = (np.round(smooth * 100) / 100).astype(np.float32)
ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                       "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
settings = os.environ["CUTS_LIST"].split(";") if os.environ.get("CUTS_LIST") else ["default", "0.25,0.5,0.75"]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    store = os.path.join(d, "s.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 24, "latitude": ny, "longitude": nx}, compress="blosc")
    size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(store) for f in fs)
    print(f"store: {arr.nbytes / 1e6:.0f} MB decoded, {size / 1e6:.0f} MB on disk (ratio {arr.nbytes / size:.2f})", flush=True)
    fn = lambda: af.dataset_from_path(store, "t2m", lon_is_360=False, device="cuda")
    os.environ["AGGFLY_HIP_GPU_DECODE"] = "1"
    fn(); torch.cuda.synchronize()
    for rep in range(2):                               # two sweeps: the order of the settings must not matter
        for sname in settings:
            os.environ.pop("AGGFLY_HIP_GPU_DECODE_CUTS", None)
            os.environ.pop("AGGFLY_HIP_GPU_DECODE_SLOTS", None)
            os.environ.pop("AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MB", None)
            if sname.startswith("tail="):                              # default cuts, another size of the host-decoded tail (MB)
                os.environ["AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MB"] = sname[5:]
            elif sname.startswith("slots="):                             # default cuts, another number of staging slots
                os.environ["AGGFLY_HIP_GPU_DECODE_SLOTS"] = sname[6:]
            elif sname != "default":
                os.environ["AGGFLY_HIP_GPU_DECODE_CUTS"] = sname
            ts = []
            for _ in range(7):
                torch.cuda.synchronize(); t0 = time.perf_counter(); got = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            assert np.array_equal(got.cube()[-3:].cpu().numpy(), arr[-3:])
            print(f"sweep {rep}  cuts {sname:<28} best {min(ts):6.2f}  median {sorted(ts)[3]:6.2f} ms", flush=True)
