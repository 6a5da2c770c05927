import os, sys, time, tempfile, numpy as np, pandas as pd, torch
sys.path.insert(0, os.getcwd())
import aggfly_amd as af
from aggfly_amd import io, synth, codec
T, ny, nx = 8760, 104, 236
arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1)
ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"), "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
d = tempfile.mkdtemp(dir="/dev/shm")
store = os.path.join(d, "s.zarr")
af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 24, "latitude": ny, "longitude": nx}, compress="blosc")
za = io.ZarrArray(os.path.join(store, "t2m"))
paths = [os.path.join(za.path, f"{i}.0.0") for i in range(365)]
out = np.empty((T, ny, nx), dtype=np.float32)
outs = [out[i*24:(i+1)*24] for i in range(365)]
gb = arr.nbytes / 1e9
for th in (1, 4, 8, 16, 32):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); codec.blosc_decode_files(paths, outs, threads=th); best = min(best, time.perf_counter() - t0)
    print("decode_files warm dst threads", th, "GB/s %.1f" % (gb / best))
t0 = time.perf_counter(); cold = np.empty((T, ny, nx), dtype=np.float32); couts = [cold[i*24:(i+1)*24] for i in range(365)]; codec.blosc_decode_files(paths, couts, threads=16); print("cold dst 16 thr GB/s %.1f" % (gb / (time.perf_counter() - t0)))
torch.cuda.synchronize()
t0 = time.perf_counter(); x = torch.from_numpy(out).cuda(); torch.cuda.synchronize(); print("H2D pageable GB/s %.1f" % (gb / (time.perf_counter() - t0)))
t0 = time.perf_counter(); x = torch.from_numpy(out).cuda(); torch.cuda.synchronize(); print("H2D pageable GB/s %.1f" % (gb / (time.perf_counter() - t0)))
for sb in (32 << 20, 64 << 20, 128 << 20, 256 << 20):
    io.zarr_to_device(store, "t2m", slab_bytes=sb)
    t0 = time.perf_counter(); io.zarr_to_device(store, "t2m", slab_bytes=sb); torch.cuda.synchronize(); print("zarr_to_device slab", sb >> 20, "MiB GB/s %.1f" % (gb / (time.perf_counter() - t0)))
import shutil; shutil.rmtree(d)
