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
        # netCDF-4 (HDF5) files, written by the real HDF5 library when an h5py interpreter is around
        h5py_python = "/opt/conda/bin/python3.9"
        if os.path.exists(h5py_python):
            import subprocess
            npy = os.path.join(d, "cube.npy")
            np.save(npy, arr)
            for tag, chunks in (("rows", (24, ny, nx)), ("tiled", (744, 52, 118))):
                nc = os.path.join(d, f"era5_{tag}.nc")
                code = ("import h5py, numpy as np; a = np.load(%r); f = h5py.File(%r, 'w', track_order=True); "
                        "t = f.create_dataset('time', data=(np.arange(a.shape[0]) + 885336).astype('i4')); t.attrs['units'] = np.bytes_('hours since 1900-01-01 00:00:00.0'); "
                        "la = f.create_dataset('latitude', data=(np.arange(a.shape[1]) * 0.25).astype('f4')); lo = f.create_dataset('longitude', data=(np.arange(a.shape[2]) * 0.25).astype('f4')); "
                        "[x.make_scale(n) for x, n in ((t, 'time'), (la, 'latitude'), (lo, 'longitude'))]; "
                        "v = f.create_dataset('t2m', data=a, chunks=%r, compression='gzip', compression_opts=1, shuffle=True); "
                        "[v.dims[i].attach_scale(x) for i, x in enumerate((t, la, lo))]; f.close()") % (npy, nc, chunks)
                if subprocess.run([h5py_python, "-c", code], capture_output=True).returncode != 0:
                    continue
                out[f"netcdf4_deflate_{tag}_ratio"] = arr.nbytes / os.path.getsize(nc)
                for name, fn in (("host_decode_then_copy", lambda: af.dataset_from_path(nc, "t2m", lon_is_360=False).to_device()),
                                 ("stream_to_hbm", lambda: af.dataset_from_path(nc, "t2m", lon_is_360=False, device="cuda"))):
                    fn(); torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter(); got = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                    assert np.array_equal(got.cube()[:5].cpu().numpy(), arr[:5]) and np.array_equal(got.cube()[-3:].cpu().numpy(), arr[-3:])
                    out[f"netcdf4_deflate_{tag}_{name}_GBps"] = gb / best
    out["host_cores"] = len(os.sched_getaffinity(0))
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/ingest_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
