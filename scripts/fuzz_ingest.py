#!/usr/bin/env python3
"""Random stores through `dataset_from_path(device="cuda")`: shapes, chunk grids (ragged edges), dtypes, codecs, Zarr formats,
shards, time windows — each read with the chunks decoded in HBM (forced, where the store allows it) and on the host threads,
and compared with the source array bit for bit.  FUZZ_LO / FUZZ_HI: the seeds."""
import os, sys, tempfile, shutil
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aggfly_amd as af
from aggfly_amd import synth

lo, hi = int(os.environ.get("FUZZ_LO", "0")), int(os.environ.get("FUZZ_HI", "120"))
fails = 0
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    T, ny, nx = int(rng.integers(5, 400)), int(rng.integers(3, 60)), int(rng.integers(3, 90))
    dtype = rng.choice([np.float32, np.float64])
    kind = rng.choice(["smooth", "noisy", "const"])
    if kind == "noisy":
        arr = synth.temperature_cube(T, ny, nx, dtype=dtype, seed=seed, scattered_nan=int(rng.integers(0, 20)))
    elif kind == "smooth":
        k = np.arange(T)[:, None, None]; y = np.arange(ny)[None, :, None]; x = np.arange(nx)[None, None, :]
        arr = (np.round((280 + 9 * np.sin(k / 37.0) + 4 * np.sin(y / 5.0) * np.cos(x / 7.0)) * 100) / 100).astype(dtype)
    else:
        arr = np.full((T, ny, nx), 3.25, dtype=dtype)
    comp = ["blosc", "blosc", "blosc", False, "zstd", "zlib"][int(rng.integers(0, 6))]
    fmt = 3 if comp == "zstd" else int(rng.choice([2, 3]))
    chunks = {"time": int(rng.integers(1, T + 1)), "latitude": int(rng.integers(1, ny + 1)), "longitude": int(rng.integers(1, nx + 1))}
    if rng.random() < 0.4:
        chunks["latitude"], chunks["longitude"] = ny, nx
    shards = None
    if fmt == 3 and rng.random() < 0.4:
        shards = {d: chunks[d] * int(rng.integers(1, 4)) for d in chunks}
    t0 = pd.Timestamp("2001-01-01")
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": pd.date_range(t0, periods=T, freq="h"),
                                                                           "latitude": 10 + 0.25 * np.arange(ny), "longitude": 200 + 0.25 * np.arange(nx)}), lon_is_360=True)
    a, b = sorted(int(v) for v in rng.integers(0, T, 2))
    sel = None if rng.random() < 0.5 or a == b else slice(t0 + pd.Timedelta(hours=a), t0 + pd.Timedelta(hours=b))
    want = arr if sel is None else arr[a:b + 1]
    d = tempfile.mkdtemp(dir=base)
    try:
        store = os.path.join(d, "s.zarr")
        af.dataset_to_zarr(ds, store, var="v", chunks=chunks, compress=comp, zarr_format=fmt, shards=shards)
        for mode in ("1", "0"):
            os.environ["AGGFLY_HIP_GPU_DECODE"] = mode
            os.environ["AGGFLY_HIP_GPU_DECODE_BATCH_MB"] = str(int(rng.choice([1, 2, 64])))
            os.environ["AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MIN_MB"] = str(int(rng.choice([0, 256])))      # 0: a host-decoded tail on these small requests too
            os.environ["AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MB"] = str(int(rng.choice([0, 1, 128])))
            got = af.dataset_from_path(store, "v", lon_is_360=True, device="cuda", time_sel=sel).cube().cpu().numpy()
            ok = got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
            if not ok:
                fails += 1
                print(f"FAILED seed {seed} mode {mode}: {T}x{ny}x{nx} {np.dtype(dtype).name} {kind} comp={comp} fmt={fmt} chunks={chunks} shards={shards} sel={a, b if sel is not None else None}", flush=True)
    except Exception as e:       # noqa: BLE001 — a fuzz run reports and goes on
        fails += 1
        print(f"FAILED seed {seed}: {type(e).__name__}: {e} | {T}x{ny}x{nx} comp={comp} fmt={fmt} chunks={chunks} shards={shards}", flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if seed % 20 == 0:
        print(f"seed {seed} failures so far: {fails}", flush=True)
print("done, failures:", fails)
