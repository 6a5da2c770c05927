"""Component rates for the converter's own layout (whole series x ~85x85 tiles, ~250 MB Blosc chunks)."""
import os, sys, time, tempfile, shutil
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.getcwd())
import aggfly_amd as af
from aggfly_amd import io, synth, codec
T, ny, nx = 8760, 104, 236
arr = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1)
ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"), "latitude": np.arange(ny) * 0.25, "longitude": np.arange(nx) * 0.25}), lon_is_360=False)
d = tempfile.mkdtemp(dir="/dev/shm")
store = os.path.join(d, "s.zarr")
af.dataset_to_zarr(ds, store, var="t2m")
za = io.ZarrArray(os.path.join(store, "t2m"))
print("chunks", za.chunks, "chunk MB", za.chunk_nbytes / 1e6)
idxs = [(0, iy, ix) for iy in range(-(-ny // za.chunks[1])) for ix in range(-(-nx // za.chunks[2]))]
locs = [za.chunk_locator(i) for i in idxs]
host = [np.empty(za.chunk_nbytes, dtype=np.uint8) for _ in idxs]
gb = sum(h.nbytes for h in host) / 1e9
for th in (1, 4, 16, 32):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); codec.decode_ranges("blosc", locs, host, threads=th); best = min(best, time.perf_counter() - t0)
    print("decode_ranges threads", th, "GB/s %.1f" % (gb / best))
pin = io._pinned_stage(4 * za.chunk_nbytes, 1)[0]
outs = [pin[i * za.chunk_nbytes:(i + 1) * za.chunk_nbytes].numpy() for i in range(4)]
t0 = time.perf_counter(); codec.decode_ranges("blosc", locs[:4], outs, threads=16); dt = time.perf_counter() - t0
print("into pinned, 4 chunks, 16 thr GB/s %.1f" % (4 * za.chunk_nbytes / 1e9 / dt))
devb = torch.empty(4 * za.chunk_nbytes, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter(); devb.copy_(pin[:4 * za.chunk_nbytes], non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("H2D pinned GB/s %.1f" % (4 * za.chunk_nbytes / 1e9 / dt))
cube = torch.empty((T, ny, nx), dtype=torch.float32, device="cuda")
tc, yc, xc = za.chunks
blk = devb[:za.chunk_nbytes].view(torch.float32).view(tc, yc, xc); torch.cuda.synchronize()
t0 = time.perf_counter(); cube[:, :yc, :xc].copy_(blk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("D2D scatter one chunk GB/s %.1f" % (za.chunk_nbytes / 1e9 / dt))
for _ in range(2):
    t0 = time.perf_counter(); io.zarr_to_device(store, "t2m"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("zarr_to_device GB/s %.1f" % (arr.nbytes / 1e9 / dt))
shutil.rmtree(d)
