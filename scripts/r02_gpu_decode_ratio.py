#!/usr/bin/env python3
"""Store -> HBM with the chunk decode on the GPU (AGGFLY_HIP_GPU_DECODE=1) and on the host threads (=0) on fields of different
compressibility: the noisy synthetic field of the benches (Blosc-LZ4 ratio ~1.2) and smooth fields quantised like real
reanalysis output (ratio 2-6).  Decoded GB/s, best of 3, stores in /dev/shm."""
import json, os, sys, tempfile, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aggfly_amd as af
from aggfly_amd import synth

YEARS = int(os.environ.get("YEARS", "1"))            # YEARS=4: a 3.4 GB store, enough batches for the pipeline's steady state
T, ny, nx = int(os.environ.get("HOURS", 8760 * YEARS)), 104, 236      # HOURS=3500: a 0.34 GB store (where the route starts by default)
k = np.arange(T)[:, None, None]; y = np.arange(ny)[None, :, None]; x = np.arange(nx)[None, None, :]
smooth = 285 + 12 * np.sin(2 * np.pi * k / 8760.0) + 5 * np.sin(2 * np.pi * (k % 24) / 24) + 8 * np.sin(y / 17.0) * np.cos(x / 23.0)
rng = np.random.default_rng(1)
fields = {"noisy (bench field)": synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15),
          "smooth + N(0, 0.3), 0.01 K steps": (np.round((smooth + rng.normal(0, 0.3, smooth.shape)) * 100) / 100).astype(np.float32),
          "smooth, 0.01 K steps": (np.round(smooth * 100) / 100).astype(np.float32)}
only = os.environ.get("FIELDS")                       # FIELDS=noisy: a substring selects the fields (with AGGFLY_HIP_INGEST_TRACE=1: the phases)
if only:
    fields = {k_: v for k_, v in fields.items() if only in k_}
elif YEARS == 1:
    fields["smooth, 0.1 K steps"] = (np.round(smooth * 10) / 10).astype(np.float32)
out = {}
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
for name, arr in fields.items():
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                           "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
    with tempfile.TemporaryDirectory(dir=base) as d:
        layouts = [("rows 24 x grid", {"time": 24, "latitude": ny, "longitude": nx}), ("tiled 744x52x118", {"time": 744, "latitude": 52, "longitude": 118})]
        if os.environ.get("CONVERTER_LAYOUT") == "1":      # what the reference's own converter writes: the whole series x 87 x 87 tiles (265 MB chunks a year)
            layouts = [("whole series x 87x87 tiles", {"time": T, "latitude": 87, "longitude": 87})]
        for lname, chunks in layouts:
            store = os.path.join(d, "s.zarr")
            af.dataset_to_zarr(ds, store, var="t2m", chunks=chunks, compress="blosc")
            size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(store) for f in fs)
            row = {"ratio": arr.nbytes / size}
            for mode in ("1", "0"):
                os.environ["AGGFLY_HIP_GPU_DECODE"] = mode
                fn = lambda: af.dataset_from_path(store, "t2m", lon_is_360=False, device="cuda")
                fn(); torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter(); got = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                assert np.array_equal(got.cube()[:3].cpu().numpy(), arr[:3]) and np.array_equal(got.cube()[-3:].cpu().numpy(), arr[-3:])
                row["gpu_decode_GBps" if mode == "1" else "host_decode_GBps"] = arr.nbytes / 1e9 / best
            out[f"{name} | {lname}"] = row
            print(name, "|", lname, row, flush=True)
            import shutil; shutil.rmtree(store)
os.makedirs("gpurun_out/r02", exist_ok=True)
if not only:
    json.dump(out, open(f"gpurun_out/r02/gpu_decode_by_ratio{'' if YEARS == 1 else '_%dyr' % YEARS}{'_%dh' % T if 'HOURS' in os.environ else ''}.json", "w"), indent=1)
