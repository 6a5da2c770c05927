#!/usr/bin/env python3
"""Which copy pattern leaves rocprofv3 --memory-copy-trace with "completion callbacks were not delivered" at exit?
MODE=plain   pinned -> device copies on a side stream, stream synchronised
MODE=event   ... whose completion another stream consumes through an event (the ingest pipeline's pattern)
MODE=slices  ... copies between SLICES of larger pinned / device buffers
MODE=ingest  a real store -> HBM read through dataset_from_path (chunks decoded in HBM)"""
import os, sys
import torch
mode = os.environ.get("MODE", "plain")
if mode == "ingest":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np, pandas as pd, tempfile
    import aggfly_amd as af
    from aggfly_amd import synth
    os.environ["AGGFLY_HIP_GPU_DECODE"] = os.environ.get("GPU_DECODE", "1")
    T, ny, nx = 24 * 30, 40, 64
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=3)
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                            "latitude": 30 + 0.25 * np.arange(ny), "longitude": 250 + 0.25 * np.arange(nx)}), lon_is_360=True)
    d = tempfile.mkdtemp()
    af.dataset_to_zarr(ds, d + "/s.zarr", var="t2m", chunks={"time": 48, "latitude": ny, "longitude": nx})
    got = af.dataset_from_path(d + "/s.zarr", "t2m", lon_is_360=True, device="cuda")
    torch.cuda.synchronize()
    print("done", bool(np.array_equal(got.cube().cpu().numpy(), cube)))
    sys.exit(0)
x = torch.ones(1 << 22).pin_memory()
dev = torch.empty(1 << 22, device="cuda")
s, w = torch.cuda.Stream(), torch.cuda.Stream()
for i in range(8):
    with torch.cuda.stream(s):
        if mode == "slices":
            dev[i << 18:(i + 1) << 18].copy_(x[i << 18:(i + 1) << 18], non_blocking=True)
        else:
            dev.copy_(x, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(s)
    if mode == "event":
        w.wait_event(ev)
        with torch.cuda.stream(w):
            dev.mul_(1.0)
s.synchronize(); w.synchronize(); torch.cuda.synchronize()
print("done", float(dev.sum()))
