#!/usr/bin/env python3
"""Host -> HBM ingestion rates: plain pageable copy, pinned double-buffered
slabs, and the Zarr decoder (raw, zlib and Blosc-LZ4 chunks; space-tiled and time-contiguous
layouts) streaming straight into HBM.  Rates are decoded (uncompressed) GB/s."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import aggfly_amd as af  # noqa: E402
from aggfly_amd import io, synth  # noqa: E402


def main():
    T, ny, nx = 8760, 104, 236          # one year of the CONUS window, f32 = 0.86 GB
    arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1)
    gb = arr.nbytes / 1e9
    out = {"bytes": arr.nbytes}
    torch.cuda.synchronize()
    for name, fn in (("pageable_copy", lambda: torch.from_numpy(arr).cuda()),):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[name + "_GBps"] = gb / dt
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"],
                                 {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                  "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None          # stores in RAM: measure decode, not the disk
    with tempfile.TemporaryDirectory(dir=base) as d:
        layouts = {"tiled": {"time": 744, "latitude": 52, "longitude": 118},      # space-tiled, like a generic store
                   "rows": {"time": 24, "latitude": ny, "longitude": nx},          # time-contiguous: whole grid per chunk
                   "auto": None}                                                    # the converter's own policy: whole series x ~85x85 tiles, ~250 MB chunks
        for comp in (False, "zlib", "blosc", "zstd"):
            for lname, chunks in layouts.items():
                if (comp is False and lname == "rows") or (lname == "auto" and comp != "blosc"):
                    continue
                store = os.path.join(d, f"s_{comp}_{lname}.zarr")
                af.dataset_to_zarr(ds, store, var="t2m", chunks=chunks, compress=comp, zarr_format=3 if comp == "zstd" else 2)
                size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(store) for f in fs)
                tag = f"zarr_{comp or 'raw'}_{lname}"
                out[tag + "_ratio"] = arr.nbytes / size
                for name, fn in (("host_decode_then_copy", lambda: af.dataset_from_path(store, "t2m", lon_is_360=False).to_device()),
                                 ("stream_to_hbm", lambda: af.dataset_from_path(store, "t2m", lon_is_360=False, device="cuda"))):
                    fn(); torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter(); got = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                    assert np.array_equal(got.cube()[:5].cpu().numpy(), arr[:5]) and np.array_equal(got.cube()[-3:].cpu().numpy(), arr[-3:])
                    out[f"{tag}_{name}_GBps"] = gb / best
    out["host_cores"] = len(os.sched_getaffinity(0))
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/ingest_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
