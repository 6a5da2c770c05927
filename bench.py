#!/usr/bin/env python3
"""Headline benchmark: grid-cell-timesteps/s of the fused transform + weighted reduce.

    python bench.py --gpus N --steps K --warmup W [--shard time|cells]      (N > 1: starts its own N ranks as a child torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (or under a launcher's environment)

Workload (BASELINE.json configs[1], shapes per SURVEY.md §8d): one year of hourly ERA5-like
2 m temperature on the 0.25 deg US-counties extent (T = 8760, 215 x 1440 = 309,600 cells,
fp64, synthetic), ~3.1k regions with area weights, and the fused plan

    dd[10,30]@date -> sum@year      +      mean@date -> power[1..4] -> sum@year      (K = 5)

A "step" is one whole pass of the hot path over one resident year: fused temporal kernel,
slot merge + shared validity, CSR weighted sums, divide, and for N > 1 the one exchange step of the
sharding.  Inputs are resident in HBM before the timed region.

  --shard time   (default) every rank owns a different year (time-axis sharding on outer-period
                 boundaries, the north_star's split): weak scaling, value = N * T * cells / max-over-ranks time;
                 the exchange is the RCCL all-gather of the region x period panel (on the collective's
                 stream beside the next step's kernels; every gather completes inside the timed region).
  --shard cells  configs[1] itself has ONE output period, so N GPUs can only split its cells: every rank
                 owns a latitude band of the SAME year, reduces it against the band's share of the weights,
                 and one RCCL all-reduce(sum) of the (K+1) x R numerators / denominators + one divide finish
                 the panel: strong scaling, value = T * cells / max-over-ranks time.

N > 1 needs one GPU per rank and runs on RCCL ("nccl").  With fewer visible GPUs than local ranks the
bench REFUSES to run unless AGGFLY_BENCH_BACKEND=gloo is set explicitly (a rehearsal: ranks share a card,
the exchange goes through the host); the JSON line always carries the backend, the visible devices and the
distinct devices the ranks really used.

The JSON line also carries
  roofline       the fused temporal kernel against the HBM peak: algorithmic bytes
                 (T * cells * 8 B per launch) / the kernel's mean duration,
                 measured with HIP events around every launch of the timed region; `traffic` is the
                 PMC-measured HBM bytes per launch from profiles/traffic.json with its provenance;
  cpu_baseline   the plain-C port of the reference's numba engine (oracle/c) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N = 1);
  other_configs  (N = 1) the other four BASELINE configs, a few steps each, kernel time and roofline;
  ingest         (N = 1) the configs[0] store read from RAM into HBM, chunks decoded in HBM / on the host threads.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes: this pool's driver only supports dmabuf IPC

HBM_PEAK_GBPS = 8000.0       # MI355X spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy rate
VALU_PEAK_LANE_INST = 256 * 4 * 16 * 2.4e9       # fp64 VALU issue peak: 256 CUs x 4 SIMDs x 16 fp64 lanes per clock x 2.4 GHz


def c2_columns():
    cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    cols += [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    return cols


def make_cube(torch, T, ny, nx, dtype, seed, steps_per_day=24, lat_lo=0.6, lat_hi=1.4, iid=False):
    """ERA5-like synthetic temperatures generated on the device, slab by slab: seasonal cycle + a cycle
    inside the day (24 hourly steps; 2 steps = (tmin, tmax) pairs; 1 = daily means) + N(0, 3).  ``iid``: 15 + N(0, 12) per element
    instead — no structure in space or time, every wave meets every branch of a data-dependent kernel (the hostile case)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    cube = torch.empty((T, ny, nx), dtype=dtype, device="cuda")
    lat = torch.linspace(lat_lo, lat_hi, ny, device="cuda", dtype=torch.float64)[None, :, None]
    slab = max(1, (1 << 28) // max(ny * nx, 1))
    for k0 in range(0, T, slab):
        k1 = min(T, k0 + slab)
        k = torch.arange(k0, k1, device="cuda", dtype=torch.float64)
        base = 15.0 + 12.0 * torch.sin(2 * np.pi * torch.floor(k / steps_per_day) / 365.0)
        if steps_per_day == 2:
            base = base + 6.0 * (2.0 * (k % 2) - 1.0)                   # tmin, tmax
        elif steps_per_day > 2:
            base = base + 6.0 * torch.sin(2 * np.pi * (k % steps_per_day) / steps_per_day)
        noise = torch.randn((k1 - k0, ny, nx), generator=g, device="cuda", dtype=torch.float32)
        if iid:
            cube[k0:k1] = (15.0 + 12.0 * noise).to(dtype)
        elif dtype == torch.float32:
            cube[k0:k1] = (base[:, None, None] * lat).to(torch.float32) + noise * 3.0
        else:
            cube[k0:k1] = base[:, None, None] * lat + noise.to(torch.float64) * 3.0
    return cube


def host_cores() -> int:
    """CPU cores this process may actually use: the affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def lib_build_id() -> str:
    """Identity of the HIP library this process runs: the first 12 hex digits of its SHA-256."""
    from aggfly_amd import hip
    h = hashlib.sha256()
    with open(hip.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:12]


def cpu_baseline(T, ny_sample, nx, seed, target_s=12.0):
    """The reference's numba-engine arithmetic (C port, OpenMP over grid rows like prange) on a
    latitude band of the same workload; every output name re-reads the raw data and every
    intermediate is materialised, exactly as the reference does (aggregate.py:133)."""
    from aggfly_amd import synth
    from oracle import cport
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    cport.build()
    ib = synth.hourly_bounds(T)
    ob = np.array([0, len(ib) - 1], dtype=np.int64)
    tab = synth.weights_table(ny_sample, nx, max(2, 3100 * ny_sample // 215), seed=7)
    ridx, cidx, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
    R = int(ridx.max()) + 1
    cube = synth.temperature_cube(T, ny_sample, nx, dtype=np.float64, seed=seed)

    def one_pass():
        outs = []
        dd = cport.resample(cube, ib, "dd", [10, 30, 0])
        outs.append(cport.resample(dd, ob, "sum"))
        for e in (1, 2, 3, 4):
            m = cport.resample(cube, ib, "mean")          # each output name restarts from the raw data
            outs.append(cport.resample(cport.power(m, e), ob, "sum"))
        x = np.stack([o.reshape(1, -1).T for o in outs])   # [K, cells, 1]
        valid = ~np.isnan(x).any(axis=0)
        den = cport.scatter_block(valid.astype(float), ridx, cidx, w, R)
        nums = [cport.scatter_block(np.where(valid, xk, 0.0), ridx, cidx, w, R) for xk in x]
        with np.errstate(invalid="ignore", divide="ignore"):
            return [np.where(den != 0, n / den, np.nan) for n in nums]

    one_pass()
    t0 = time.perf_counter()
    reps = 0
    while True:
        one_pass()
        reps += 1
        if time.perf_counter() - t0 > target_s or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    cores = min(cores, ny_sample)          # OpenMP runs over grid rows, like numba's prange
    return {"value": T * ny_sample * nx / dt, "unit": "grid-cell-timesteps/s", "cores": cores, "kind": "port",
            "sample": f"{ny_sample}x{nx} latitude band of the workload grid, T={T}, fp64, {reps} passes of {dt:.2f} s "
                      f"(C/OpenMP port of the reference's numba engine, oracle/c: `omp parallel for` over the band's {ny_sample} grid rows like "
                      "prange(NY), each cell walking time with the reference's stride of a whole grid row, the raw data re-read once per output "
                      f"name: 5 names -> {5 * T * ny_sample * nx / dt / cores / 1e6:.0f} M raw elements/s per core; the reference's published numba run — 4 names, "
                      "15.2 s for 8760x721x1440 on 32 cores, internal/backend-plan.md:4 — works out to 75 M per core, so the port stands in for numba's "
                      "speed as well as its arithmetic.  It is slower than the numpy restatement of the dask path only because that restatement "
                      "carries none of dask's scheduling overhead: the reference's own dask engine is 12x SLOWER than its numba engine)"}


def cpu_baseline_dask_path(T, ny_sample, nx, seed, target_s=8.0):
    """The reference's dask-path arithmetic (vectorised numpy reducers called once per resample
    group, `aggfly/aggregate/temporal.py:236-239,266-311`; `np.power`; `np.add.at` scatter) with a
    thread pool over blocks of days standing in for the dask threaded scheduler."""
    from concurrent.futures import ThreadPoolExecutor
    from aggfly_amd import synth
    from oracle import ref_temporal as rt
    from oracle.ref_spatial import scatter_block
    cores = host_cores()
    ib = synth.hourly_bounds(T)
    ndays = len(ib) - 1
    tab = synth.weights_table(ny_sample, nx, max(2, 3100 * ny_sample // 215), seed=7)
    ridx, cidx, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
    R = int(ridx.max()) + 1
    cube = synth.temperature_cube(T, ny_sample, nx, dtype=np.float64, seed=seed)
    blocks = np.linspace(0, ndays, min(ndays, cores * 4) + 1).astype(int)

    def temporal(calc, ddargs=None):
        def work(i):
            g0, g1 = blocks[i], blocks[i + 1]
            return rt.dask_resample(cube[ib[g0]:ib[g1]], ib[g0:g1 + 1] - ib[g0], calc, ddargs)
        with ThreadPoolExecutor(max_workers=cores) as ex:
            return np.concatenate(list(ex.map(work, range(len(blocks) - 1))))

    def one_pass():
        outs = [temporal("dd", [10, 30, 0]).sum(axis=0)]
        for e in (1, 2, 3, 4):
            outs.append(np.power(temporal("mean"), e).sum(axis=0))      # every output name restarts from the raw data
        x = np.stack([o.reshape(-1, 1) for o in outs])
        valid = ~np.isnan(x).any(axis=0)
        den = scatter_block(valid.astype(float), ridx, cidx, w, R)
        return [scatter_block(np.where(valid, xk, 0.0), ridx, cidx, w, R) / den for xk in x]

    with np.errstate(invalid="ignore", divide="ignore"):
        one_pass()
        t0 = time.perf_counter()
        reps = 0
        while True:
            one_pass()
            reps += 1
            if time.perf_counter() - t0 > target_s or reps >= 20:
                break
    dt = (time.perf_counter() - t0) / reps
    return {"value": T * ny_sample * nx / dt, "unit": "grid-cell-timesteps/s", "cores": cores, "kind": "port",
            "sample": f"{ny_sample}x{nx} band, T={T}, fp64, {reps} passes of {dt:.2f} s (numpy restatement of the reference's dask path, "
                      f"thread pool of {cores})"}


# ---------------------------------------------------------------------------------------------------------------------
# the other BASELINE configs (N = 1): a few steps each, kernel time by HIP events, whole pass by the host clock
# ---------------------------------------------------------------------------------------------------------------------
def _valu_counts():
    """VALU instructions per grid-cell-timestep of the VALU-bound kernels, from the committed PMC passes (profiles/)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "valu_counts.json")))
    except (OSError, ValueError):
        return {}


def _cpu_dask_path_panel(cube, ib, ob, cols, ridx, cidx, w, R, cores):
    """One whole pass of the reference's dask-path arithmetic over a host cube (see `cpu_baseline_dask_path`): per output name the raw data
    is re-read, reduced per resample group by the vectorised numpy reducers (thread pool over blocks of days = the threaded scheduler),
    transformed, reduced per output period; then shared validity, `np.add.at` scatter, divide.  -> res[K, R, P]"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ref_temporal as rt
    from oracle.ref_spatial import scatter_block
    ndays = len(ib) - 1
    blocks = np.linspace(0, ndays, min(ndays, cores * 4) + 1).astype(int)

    def temporal(calc, ddargs=None):
        def work(i):
            g0, g1 = blocks[i], blocks[i + 1]
            return rt.dask_resample(cube[ib[g0]:ib[g1]], ib[g0:g1 + 1] - ib[g0], calc, ddargs)
        with ThreadPoolExecutor(max_workers=cores) as ex:
            return np.concatenate(list(ex.map(work, range(len(blocks) - 1))))

    outs = []
    for c in cols:
        x = temporal(c["inner"], list(c["inner_args"]) if "inner_args" in c else None)     # every output name restarts from the raw data
        if c.get("transform") == "pow":
            x = np.power(x, c["transform_arg"])
        outs.append(np.stack([x[ob[p]:ob[p + 1]].sum(axis=0) for p in range(len(ob) - 1)], axis=-1).reshape(-1, len(ob) - 1))
    x = np.stack(outs)                                                                       # [K, cells, P]
    valid = ~np.isnan(x).any(axis=0)
    den = scatter_block(valid.astype(float), ridx, cidx, w, R)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.stack([scatter_block(np.where(valid, xk, 0.0), ridx, cidx, w, R) / den for xk in x])


def run_store_jobs(torch, cpu=True):
    """SURVEY §8d (ii) and §8f N2 beside the kernels, on the BASELINE configs[0] store (one year of hourly f32 on 104 x 236 cells, Blosc-LZ4 +
    shuffle, 24-step chunks, written to RAM so that no disk is measured):
      ingest      store -> HBM through `dataset_from_path(device="cuda")`, chunks decoded in HBM / on the host threads: decoded GB/s
      end_to_end  store -> `dataset_from_path` (+ K -> degC in HBM) -> `weights_from_objects` -> `aggregate_dataset` -> the region x period
                  DataFrame, for the configs[0] spec (mean@date -> power[1,2] -> sum@year) and the configs[1] spec (dd[10,30] + power[1..4]) on
                  that store, both decode routes: grid-cell-timesteps/s of the whole job, best of 3 after a warm job (the warm job builds
                  the plan, uploads the weights and page-locks the staging buffers: a long-running service's state);
                  `cpu`: the same two jobs on this box's host cores — the store read by the same native chunk codec into RAM, then the
                  faster restatement of the reference's CPU engines (its dask path, numpy + thread pool: `cpu_baseline`'s winner) — the
                  oracle, as the checker of nothing here: a stated baseline.
    The GPU side is the product path only."""
    import shutil, tempfile
    import pandas as pd
    import aggfly_amd as af
    from aggfly_amd import synth
    T, ny, nx, R = 8760, 104, 236, 3100
    k = np.arange(T, dtype=np.float32)[:, None, None]
    yy, xx = np.arange(ny, dtype=np.float32)[None, :, None], np.arange(nx, dtype=np.float32)[None, None, :]
    fields = {      # how well a field compresses decides how many bytes cross PCIe: the noisy bench field and one quantised like reanalysis output
        "bench field (N(0, 3) noise in the mantissa)": lambda: synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15),
        "smooth field in 0.01 K steps": lambda: (np.round((285 + 12 * np.sin(2 * np.pi * k / 8760) + 5 * np.sin(2 * np.pi * (k % 24) / 24)
                                                            + 8 * np.sin(yy / 17) * np.cos(xx / 23)) * 100) / 100).astype(np.float32)}
    specs = {"configs[0]: mean@date -> power[1,2] -> sum@year (K=2)":
                 dict(tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                            ("aggregate", {"calc": "sum", "groupby": "year"})]),
             "configs[1]: dd[10,30]@date -> sum@year + mean@date -> power[1..4] -> sum@year (K=5)":
                 dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "year"})],
                      tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 5)}),
                            ("aggregate", {"calc": "sum", "groupby": "year"})])}
    cpu_cols = {0: [dict(inner="mean", transform="pow", transform_arg=e) for e in (1, 2)],
                1: [dict(inner="dd", inner_args=(10, 30, 0))] + [dict(inner="mean", transform="pow", transform_arg=e) for e in (1, 2, 3, 4)]}
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    ingest = {"workload": "BASELINE configs[0] store -> HBM: T=8760 hourly f32, 104x236 cells, Zarr v2, Blosc-LZ4 + shuffle, 365 chunks of 24 steps, store in RAM",
              "unit": "GB/s decoded", "fields": {}}
    e2e = {"workload": "store (RAM; Zarr v2, Blosc-LZ4 + shuffle, 365 chunks of 24 steps; T=8760 hourly f32 in K, 104x236 cells) -> dataset_from_path (+ K -> degC) -> "
                       "weights_from_objects (3100 regions) -> aggregate_dataset -> DataFrame; whole-job grid-cell-timesteps/s, best of 3 after a warm job",
           "unit": "grid-cell-timesteps/s", "cell_steps": T * ny * nx, "jobs": {}}
    saved = os.environ.get("AGGFLY_HIP_GPU_DECODE")
    tindex = pd.date_range("2001-01-01", periods=T, freq="h")
    lat, lon = 25 + 0.25 * np.arange(ny), 235 + 0.25 * np.arange(nx)
    tab = synth.weights_table(ny, nx, R, seed=2)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i:05d}" for i in range(int(tab.index_right.max()) + 1)]}))
    routes = (("chunks_decoded_in_hbm", "1"), ("chunks_decoded_on_host_threads", "0"))
    try:
        for fi, (name, make) in enumerate(fields.items()):
            arr = make()
            ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": tindex, "latitude": lat, "longitude": lon}), lon_is_360=True)
            store = os.path.join(d, "c0.zarr")
            af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 24, "latitude": ny, "longitude": nx}, compress="blosc")
            size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(store) for f in fs)
            ent = {"decoded_bytes": int(arr.nbytes), "store_bytes": size, "blosc_ratio": arr.nbytes / size}
            for key, mode in routes:
                os.environ["AGGFLY_HIP_GPU_DECODE"] = mode
                fn = lambda: af.dataset_from_path(store, "t2m", lon_is_360=True, device="cuda")
                got = fn(); torch.cuda.synchronize()
                ok = bool(np.array_equal(got.cube()[:2].cpu().numpy(), arr[:2]) and np.array_equal(got.cube()[-2:].cpu().numpy(), arr[-2:]))
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter(); got = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                ent[key] = {"GBps": arr.nbytes / 1e9 / best, "ms": best * 1e3, "first_and_last_steps_equal_the_source": ok}
                del got
            ingest["fields"][name] = ent
            if fi == 0:      # the whole jobs, on the bench field's store
                e2e["store_bytes"], e2e["blosc_ratio"] = size, arr.nbytes / size
                frames = {}
                for si, (sname, spec) in enumerate(specs.items()):
                    job = {}
                    for key, mode in routes:
                        os.environ["AGGFLY_HIP_GPU_DECODE"] = mode

                        def run_job():
                            t = [time.perf_counter()]
                            dsd = af.dataset_from_path(store, "t2m", lon_is_360=True, preprocess=lambda x: x - 273.15, device="cuda")
                            torch.cuda.synchronize(); t.append(time.perf_counter())
                            wts = af.weights_from_objects(dsd, gr, table=tab)
                            df = af.aggregate_dataset(dataset=dsd, weights=wts, **spec)
                            torch.cuda.synchronize(); t.append(time.perf_counter())
                            return df, np.diff(t)
                        run_job()
                        best = None
                        for _ in range(3):
                            df, dt = run_job()
                            if best is None or dt.sum() < best.sum():
                                best = dt
                        job[key] = {"value": T * ny * nx / float(best.sum()), "total_ms": float(best.sum()) * 1e3, "open_decode_h2d_ms": float(best[0]) * 1e3,
                                    "weights_aggregate_frame_ms": float(best[1]) * 1e3, "rows": int(len(df)), "columns": [c for c in df.columns if c not in ("geoid", "time")]}
                        frames[si] = df
                    e2e["jobs"][sname] = job
                if cpu:
                    try:
                        cores = host_cores()
                        ib = synth.hourly_bounds(T)
                        ob = np.array([0, len(ib) - 1], dtype=np.int64)
                        ridx, cidx, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
                        nR = int(ridx.max()) + 1
                        for si, sname in enumerate(specs):
                            best, res = None, None
                            for _ in range(2):
                                t0 = time.perf_counter()
                                hd = af.dataset_from_path(store, "t2m", lon_is_360=True)                  # host read: the native chunk codec on the host threads
                                cube = np.asarray(hd.cube(), dtype=np.float32) - np.float32(273.15)
                                t1 = time.perf_counter()
                                res = _cpu_dask_path_panel(cube, ib, ob, cpu_cols[si], ridx, cidx, w, nR, cores)
                                frame = pd.DataFrame({"geoid": gr.shp["geoid"].to_numpy(), **{f"c{j}": res[j, :, 0] for j in range(len(res))}})
                                dt = (time.perf_counter() - t0, t1 - t0)
                                if best is None or dt[0] < best[0]:
                                    best = dt
                            cols_gpu = e2e["jobs"][sname]["chunks_decoded_in_hbm"]["columns"]
                            gpu = frames[si][cols_gpu].to_numpy()
                            # (the DataFrame's columns are alphabetical — dd before tavg_* — like the restatement's column list)
                            agree = bool(np.allclose(gpu, res[:, :, 0].T[:len(gpu)], rtol=2e-5, equal_nan=True)) if gpu.shape == res[:, :, 0].T.shape else None
                            e2e["jobs"][sname]["cpu"] = {"value": T * ny * nx / best[0], "total_ms": best[0] * 1e3, "read_decode_ms": best[1] * 1e3, "cores": cores, "kind": "port",
                                                         "sample": "the whole job, best of 2: the store read by the native chunk codec into RAM, then the numpy restatement of the "
                                                                   "reference's dask path (thread pool of %d; float32 data as stored, so it agrees with the GPU's float64 accumulation to ~1e-5)" % cores,
                                                         "panel_agrees_with_gpu_to_2e-5": agree}
                    except Exception as e:      # the baseline must never take the GPU figures down with it
                        e2e["cpu_failed"] = repr(e)
            shutil.rmtree(store, ignore_errors=True)
    finally:
        if saved is None:
            os.environ.pop("AGGFLY_HIP_GPU_DECODE", None)
        else:
            os.environ["AGGFLY_HIP_GPU_DECODE"] = saved
        shutil.rmtree(d, ignore_errors=True)
    return ingest, e2e


def run_other_configs(torch, steps=10, warmup=10):      # the first ~10 launches after idle run up to 15 % slow (clock ramp: scripts/probe/back_to_back.py)
    from aggfly_amd import hip, synth
    out = []
    valu = _valu_counts()
    edges = np.arange(-20, 50, 5.0)
    cfgs = [
        dict(name="C1", workload="BASELINE configs[0] on the counties extent: hourly t2m 1 year, 215x1440 cells, 3100 regions area weights, "
                                 "mean@date->power[1,2]->sum@year, f32 storage, K=2",
             T=8760, ny=215, nx=1440, spd=24, periods=1, R=3100, dtype="f32", secondary=False,
             cols=[dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2)]),
        dict(name="C3", workload="BASELINE configs[2] shape: hourly t2m 40 years (T=350640), CONUS window 104x236 cells, 3100 regions with "
                                 "population secondary weights, mean@date->power[1,2]->sum@year, f32 storage, K=2, P=40",
             T=350640, ny=104, nx=236, spd=24, periods=40, R=3100, dtype="f32", secondary=True,
             cols=[dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2)]),
        dict(name="C4", workload="BASELINE configs[3] shape: daily tas 251 years noleap (T=91615), 180x288 cells, 3600 regions with cropland "
                                 "secondary weights, 13 temperature bins (5 degC, -20..45) per year, f32 storage, K=13, P=251",
             T=91615, ny=180, nx=288, spd=1, periods=251, R=3600, dtype="f32", secondary=True, single_level=True,
             cols=[dict(inner="bins", inner_args=(edges[i], edges[i + 1], 0)) for i in range(13)]),
        dict(name="C5", workload="BASELINE configs[4] shape: 0.1 deg global 1801x3600 cells, 365 (tmin, tmax) pairs (T=730), 40000 regions, "
                                 "sine_dd[10,30]@date->sum@year, f32 storage, K=1",
             T=730, ny=1801, nx=3600, spd=2, periods=1, R=40000, dtype="f32", secondary=False,
             cols=[dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum")]),
        # the same shape on a field without any structure: the sine_dd kernel is data-dependent (which lanes of a wave sit inside a
        # threshold's window decides how many arcs the wave evaluates), and this is its worst case — labelled as such
        dict(name="C5_iid", workload="configs[4]'s shape and plan on an IID cube (15 + N(0, 12) per element: neighbouring cells and days unrelated) — the "
                                     "hostile case of the data-dependent sine_dd kernel; the C5 row above is the SURVEY 8d field",
             T=730, ny=1801, nx=3600, spd=2, periods=1, R=40000, dtype="f32", secondary=False, iid=True,
             cols=[dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum")]),
        # many output periods: the configs[1] columns as a DAILY panel (`groupby: date` as the last step: aggfly/aggregate/temporal.py:441-456,
        # spatial.py:110-133) — 365 periods x 5 columns x 3100 regions from one year of hourly float32
        dict(name="DAILY", workload="daily panel: hourly t2m 1 year f32 storage, 215x1440 cells, 3100 regions, dd[10,30]@date + mean@date->power[1..4] with NO "
                                    "second level (one output row per region and day: P=365, K=5); the weighted sums per region are formed inside the streaming "
                                    "kernel at every period end",
             T=8760, ny=215, nx=1440, spd=24, periods=365, R=3100, dtype="f32", secondary=False,
             cols=[dict(inner="dd", inner_args=(10, 30, 0))] + [dict(inner="mean", transform="pow", transform_arg=e) for e in (1, 2, 3, 4)]),
        # the ONE workload the reference published a number for (/root/reference/benchmarks/bench_engine.py:19-23,60-69 ->
        # internal/backend-plan.md:4-5): temporal stage of mean@date -> power[1..4] -> sum@month on one year of GLOBAL 0.25 deg
        # hourly float32.  Here the whole path runs (the spatial stage too), on a synthetic field of that shape.
        dict(name="REF", workload="the reference's own published benchmark shape (benchmarks/bench_engine.py): hourly float32, 1 year, GLOBAL 0.25 deg "
                                  "721x1440 cells (36.4 GB resident), mean@date->power[1..4]->sum@month, K=4, P=12, + 3100 regions here; published for "
                                  "the temporal stage alone on a 32-core CPU: 15.2 s (numba engine) / 179 s (dask engine), internal/backend-plan.md:4-5",
             T=8760, ny=721, nx=1440, spd=24, periods=12, months=True, R=3100, dtype="f32", secondary=False,
             cols=[dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]),
    ]
    only = os.environ.get("AGGFLY_BENCH_ONLY")          # e.g. "C5": one config (PMC passes over a single kernel)
    for c in cfgs:
        if only and c["name"] not in only.split(","):
            continue
        t_cfg = time.perf_counter()
        try:
            T, ny, nx = c["T"], c["ny"], c["nx"]
            C = ny * nx
            elem = 4 if c["dtype"] == "f32" else 8
            dt_t = torch.float32 if c["dtype"] == "f32" else torch.float64
            cube = make_cube(torch, T, ny, nx, dt_t, seed=20260105, steps_per_day=c["spd"], iid=bool(c.get("iid")))
            ib = synth.hourly_bounds(T, c["spd"])
            G1 = len(ib) - 1
            ob = np.round(np.linspace(0, G1, c["periods"] + 1)).astype(np.int64)
            if c.get("months"):                                         # calendar months of a 365-day year
                ob = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])]).astype(np.int64)
            if c.get("single_level"):
                ib, ob = ib[ob], np.arange(c["periods"] + 1, dtype=np.int64)
            tab = synth.weights_table(ny, nx, c["R"], seed=7, secondary=c["secondary"])
            R = int(tab["index_right"].max()) + 1
            csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, C)
            plan = hip.FusedPlan(T, C, hip.F32 if c["dtype"] == "f32" else hip.F64, ib, ob, c["cols"])
            outb = plan.run(cube, csr)
            for _ in range(warmup):
                plan.run(cube, csr, out=outb)
            torch.cuda.synchronize()
            plan.profile_begin(steps)
            t0 = time.perf_counter()
            for _ in range(steps):
                plan.run(cube, csr, out=outb)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            kms = plan.profile_end()
            k_ms = float(np.mean(kms))
            hbm = T * C * elem / (k_ms * 1e-3) / 1e9
            desc = plan.describe()
            route = ("region-fused period ends (weighted sums per region inside the temporal kernel) + k_rf_reduce" if "last-run=region-fused" in desc
                     else ("packed counts gathered directly" if "packed-counts" in desc else "period partials + k_csr_spmm_slots"))
            row = {"config": c["name"], "workload": c["workload"], "dtype": c["dtype"], "steps": steps,
                   "ms_per_step": dt / steps * 1e3, "value": T * C * steps / dt,
                   "kernel": desc.split()[0].replace("variant=", "") + ("_rf" if "last-run=region-fused" in desc else ""), "spatial_route": route,
                   "kernel_ms_mean": k_ms, "bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBPS,
                   "algorithmic_bytes_per_launch": T * C * elem, "regions": R, "nnz": int(csr.nnz)}
            try:      # this box's bare read of the very cube the kernel just streamed (afhip_read_probe: 8 bytes per lane, single-wave workgroups, four
                # rows in flight, no arithmetic — the arm tuned on the headline shape; a kernel whose launch suits its shape better can pass it)
                pm = hip.read_probe(cube, 6)
                bare = T * C * elem / (float(np.median(pm[1:])) * 1e-3) / 1e9
                row.update({"bare_read_GBps": bare, "hbm_vs_bare_read": hbm / bare})
            except Exception as e:      # a measuring aid must never take the row down with it
                row["read_probe_error"] = f"{type(e).__name__}: {e}"
            try:      # PMC-measured HBM bytes per launch of this shape's kernel, where a counter pass over bench.py itself has been committed
                te = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"{c['name'].lower()}_{c['dtype']}_T{T}_C{C}")
                if te:
                    row.update({"traffic": te.get("hbm_bytes_per_launch"), "traffic_build": te.get("build"), "traffic_source": te.get("source"),
                                "traffic_measured_on_this_build": te.get("build") == lib_build_id()})
            except (OSError, ValueError):
                pass
            vc = valu.get(c["name"])
            if vc:      # a VALU-bound kernel: its own roofline beside the HBM figure
                inst = vc["valu_inst_per_cell_step"] * T * C / (k_ms * 1e-3)
                row.update({"bound": "valu", "achieved": inst, "peak": VALU_PEAK_LANE_INST, "unit": "lane-instructions/s",
                            "frac": inst / VALU_PEAK_LANE_INST, "valu_inst_per_cell_step": vc["valu_inst_per_cell_step"],
                            "valu_count_source": vc.get("source"), "hbm_achieved_GBps": hbm, "hbm_frac": hbm / HBM_PEAK_GBPS})
            out.append(row)
            del cube, csr, plan, outb
        except Exception as e:  # never take the headline number down
            out.append({"config": c["name"], "workload": c["workload"], "error": f"{type(e).__name__}: {e}"})
        torch.cuda.empty_cache()
        out[-1]["setup_and_run_s"] = round(time.perf_counter() - t_cfg, 2)
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without torchrun's environment: start the N ranks ourselves, one process per GPU, as a
    CHILD `python -m torch.distributed.run` (never an exec: nothing here has touched the GPU, and the parent only waits),
    pass its output through and return its exit code.  This is the parallel form of the reference's sequential per-year loop
    (`aggfly/cli/pipeline.py:138-149`): rank r takes year r.  With fewer GPUs than ranks the launch is refused here, in one
    line, unless AGGFLY_BENCH_BACKEND=gloo asks for a rehearsal.  AGGFLY_BENCH_DRY_LAUNCH=1 prints the child command as JSON
    and starts nothing (tests)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    if os.environ.get("AGGFLY_BENCH_DRY_LAUNCH"):
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    if os.environ.get("AGGFLY_BENCH_BACKEND") != "gloo":
        import torch          # counting devices does not initialise the GPU
        ndev = torch.cuda.device_count()
        if ndev < n:
            print(f"bench.py: --gpus {n} but only {ndev} GPU(s) visible. RCCL needs one GPU per rank; a number measured with ranks sharing a "
                  "card is not a scaling result. Set AGGFLY_BENCH_BACKEND=gloo to rehearse the N > 1 code path on fewer cards.", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # 100 x 3.2 ms: a timed region of ~0.3 s
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ny", type=int, default=215)
    ap.add_argument("--nx", type=int, default=1440)
    ap.add_argument("--T", type=int, default=8760)
    ap.add_argument("--regions", type=int, default=3100)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--shard", default="time", choices=["time", "cells"],
                    help="N > 1: time = one year per GPU + all-gather (weak scaling); cells = latitude bands of one year + all-reduce (strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-ingest", action="store_true", help="skip the store -> HBM figure (N = 1, after other_configs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from aggfly_amd import distributed as D, hip, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE") or world)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")
    ndev = torch.cuda.device_count()
    backend = None
    if world > 1:
        forced = os.environ.get("AGGFLY_BENCH_BACKEND")
        if ndev < local_world and forced != "gloo":
            raise SystemExit(f"bench.py: {local_world} ranks on this node but only {ndev} GPU(s) visible. RCCL needs one GPU per rank; "
                             "a number measured with ranks sharing a card is not a scaling result. Set AGGFLY_BENCH_BACKEND=gloo "
                             "to rehearse the N > 1 code path on fewer cards (the line then says backend gloo).")
        backend = forced or "nccl"
    dev = local_rank % max(ndev, 1)          # only a gloo rehearsal ever shares a card
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    hip.require_gpu()

    T, ny, nx = args.T, args.ny, args.nx
    C = ny * nx
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    elem = 8 if args.dtype == "f64" else 4
    cells_mode = args.shard == "cells"
    ib = synth.hourly_bounds(T)
    ob = np.array([0, len(ib) - 1], dtype=np.int64)
    tab = synth.weights_table(ny, nx, args.regions, seed=7)
    R = int(tab["index_right"].max()) + 1
    rows_, cols_, w_ = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
    if cells_mode:
        # strong scaling: rank r owns latitude rows [y0, y1) of the same year and the weights that fall on them
        y0, y1 = D.split_even(ny, rank, world)
        if y1 <= y0:
            raise SystemExit(f"--shard cells: {world} ranks for {ny} latitude rows")
        my_ny = y1 - y0
        cube = make_cube(torch, T, my_ny, nx, dtype, seed=20260101 + rank,
                         lat_lo=0.6 + 0.8 * y0 / max(ny - 1, 1), lat_hi=0.6 + 0.8 * (y1 - 1) / max(ny - 1, 1))
        br, bc, bw = D.band_csr_triplets(rows_, cols_, w_, ny, nx, y0, y1)
        csr = hip.CSR(br, bc, bw, R, my_ny * nx)
        my_C = my_ny * nx
    else:
        cube = make_cube(torch, T, ny, nx, dtype, seed=20260101 + rank)          # rank r owns year r
        csr = hip.CSR(rows_, cols_, w_, R, C)
        my_C = C
    cols = c2_columns()
    K = len(cols)
    plan = hip.FusedPlan(T, my_C, hip.F64 if args.dtype == "f64" else hip.F32, ib, ob, cols)
    out = {"num": torch.empty((K, R, 1), dtype=torch.float64, device="cuda"),
           "den": torch.empty((R, 1), dtype=torch.float64, device="cuda"),
           "res": torch.empty((K, R, 1), dtype=torch.float64, device="cuda")}
    # time sharding, N > 1: the region x period panel of every step is all-gathered.  Two result buffers, so that
    # the gather of step i (on the collective's own stream) overlaps the kernels of step i + 1; a buffer is reused
    # only after its gather has finished, and every gather is waited for inside the timed region.
    pipelined = world > 1 and not cells_mode and os.environ.get("AGGFLY_BENCH_PIPELINE", "1") != "0"
    outs = [out] + ([{k: torch.empty_like(v) for k, v in out.items()}] if pipelined else [])
    use_flat = world > 1 and dist.get_backend() == "nccl"
    gathered, flat = None, None
    if world > 1 and not cells_mode:
        if use_flat:        # one RCCL call straight into [world, K, R, 1]
            gathered = [torch.empty((world,) + tuple(out["res"].shape), dtype=torch.float64, device="cuda") for _ in outs]
        else:
            gathered = [[torch.empty_like(out["res"]) for _ in range(world)] for _ in outs]
    if cells_mode:
        flat = torch.empty(((K + 1) * R,), dtype=torch.float64, device="cuda")     # num | den of one step, summed over ranks
        full = torch.empty((K, R, 1), dtype=torch.float64, device="cuda")
    pending = [None] * len(outs)

    def step(i):
        b = i % len(outs)
        if pending[b] is not None:
            pending[b].wait()                                 # the gather that last read this buffer
            pending[b] = None
        plan.run(cube, csr, out=outs[b])
        if cells_mode:
            # one exchange: sum the ranks' numerators and denominators, then the library's divide (spatial.py:127-133)
            flat[:K * R].copy_(outs[b]["num"].reshape(-1))
            flat[K * R:].copy_(outs[b]["den"].reshape(-1))
            if world > 1:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            hip.panel_divide(flat[:K * R].view(K, R, 1), flat[K * R:].view(R, 1), out=full)
        elif world > 1:
            if use_flat:
                w = dist.all_gather_into_tensor(gathered[b], outs[b]["res"], async_op=pipelined)
            else:
                w = dist.all_gather(gathered[b], outs[b]["res"], async_op=pipelined)
            pending[b] = w if pipelined else None

    def drain():
        for b in range(len(outs)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    plan.profile_begin(args.steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = plan.profile_end()
    # outside the timed region: this box's streaming-read ceiling for the very cube the kernel just read — ten back-to-back launches of
    # the library's probe kernel (the temporal kernel's access pattern, no arithmetic: afhip_read_probe) by HIP events
    probe_ms = []
    if rank == 0:
        try:
            probe_ms = hip.read_probe(cube, 10)
        except Exception as e:      # a measuring aid must never take the headline down with it
            print(f"bench.py: read probe failed: {e}", file=sys.stderr)
    # the exchanged panel really holds this rank's result of the last step
    b = (args.steps - 1) % len(outs)
    if world > 1 and not cells_mode:
        mine = gathered[b][rank]
        if not torch.equal(torch.nan_to_num(mine), torch.nan_to_num(outs[b]["res"])):
            raise SystemExit(f"rank {rank}: gathered panel differs from the local result")
    if cells_mode and world == 1:
        if not torch.equal(torch.nan_to_num(full), torch.nan_to_num(outs[b]["res"])):
            raise SystemExit("cell arm: the divided panel differs from the plan's own result")
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # which devices did the ranks really run on?
    props = torch.cuda.get_device_properties(dev)
    ident = (os.uname().nodename, str(getattr(props, "uuid", "")) or f"index{dev}", props.name)
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)

    if rank == 0:
        build = lib_build_id()
        traffic, traffic_source, traffic_build = None, None, None
        try:   # PMC-derived HBM bytes per launch for this workload, collected in a separate rocprofv3 --pmc pass
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get(f"c2_{args.dtype}_T{T}_C{my_C}", {})
            traffic = ent.get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_source = "profiles/traffic.json: " + str(ent.get("source"))
                traffic_build = ent.get("build")
        except (OSError, ValueError):
            pass
        ms_per_step = dt / args.steps * 1e3
        units = (T * C) if cells_mode else (world * T * C)          # cells: one year split over the ranks; time: one year per rank
        value = units * args.steps / dt
        k_ms = float(np.mean(kms)) if kms else float("nan")
        achieved = T * my_C * elem / (k_ms * 1e-3) / 1e9
        if world == 1:
            sharding = "single GPU" + (" (cell arm: num/den -> library divide, no exchange)" if cells_mode else "")
        elif cells_mode:
            sharding = (f"cell axis: {world} latitude bands of one year, one {'RCCL' if backend == 'nccl' else backend} all-reduce(sum) of "
                        f"(K+1) x R doubles + one divide per step")
        else:
            sharding = (f"time axis, one year per GPU; {'RCCL' if backend == 'nccl' else backend} all-gather of the panel"
                        + (", overlapped with the next step's kernels" if pipelined else ""))
        line = {
            "metric": "grid-cell-timesteps/s (fused transform+weighted reduce)",
            "value": value, "unit": "grid-cell-timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if cells_mode else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "backend": backend if world > 1 else "none (single process)", "ranks": world, "devices_visible": ndev,
            "devices_used": len({i[:2] for i in idents}), "device_names": sorted({i[2] for i in idents}),
            "config": {"workload": "BASELINE configs[1]: ERA5-like hourly t2m 0.25deg, 1 year" + ("" if cells_mode else " per GPU") + ", US-counties extent "
                                   f"{ny}x{nx} cells, {R} regions area weights, fused dd[10,30]@date->sum@year + "
                                   "mean@date->power[1..4]->sum@year, %s, K=5" % ("fp64" if args.dtype == "f64" else "fp32 storage / fp64 accumulation"),
                       "T": T, "cells": C, "cells_per_rank": my_C, "regions": R, "columns": K, "nnz": int(csr.nnz),
                       "shard": args.shard, "sharding": sharding, "plan": plan.describe()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_source, "traffic_build": traffic_build, "build": build,
                         "traffic_measured_on_this_build": bool(traffic_build) and traffic_build == build,
                         "kernel": "k_fused_temporal", "kernel_ms_mean": k_ms, "launches": len(kms),
                         # the spread of the per-launch HIP-event times: a run that slows down as it goes (power / thermal) shows in
                         # first10 vs last10, a process that sits in a slow mode from its first launch does not
                         "kernel_ms": ({"min": float(np.min(kms)), "median": float(np.median(kms)), "max": float(np.max(kms)),
                                        "first10_mean": float(np.mean(kms[:10])), "last10_mean": float(np.mean(kms[-10:]))} if kms else None),
                         "algorithmic_bytes_per_launch": T * my_C * elem},
        }
        if probe_ms:
            # "a measured device-copy/triad bandwidth on the box ... state which denominator is used" — `frac` stays on the
            # 8 TB/s spec peak; `frac_of_measured` is the same kernel time against what a bare read of the same cube reached in this process
            ceiling = T * my_C * elem / (float(np.median(probe_ms)) * 1e-3) / 1e9
            line["roofline"].update({"measured_read_ceiling_GBps": ceiling, "frac_of_measured": achieved / ceiling,
                                     "read_ceiling": {"kernel": "k_read_probe (8 B per lane, single-wave workgroups, 4 nt row loads in flight, no arithmetic)",
                                                      "launches": len(probe_ms), "ms_median": float(np.median(probe_ms)), "ms_min": float(np.min(probe_ms)),
                                                      "frac_of_spec_peak": ceiling / HBM_PEAK_GBPS, "denominator_of_frac": "8000 GB/s spec peak"}})
        if world > 1 and len({i[:2] for i in idents}) < world:
            line["warning"] = "ranks shared a GPU (gloo rehearsal): not a scaling measurement"
        if world == 1 and not args.no_cpu_baseline:
            # both CPU engines of the reference, restated (oracle/): the faster one is THE baseline
            cands = []
            for fn in (cpu_baseline, cpu_baseline_dask_path):
                try:
                    cands.append(fn(T, max(8, min(host_cores(), 64)), nx, seed=20260101))
                except Exception as e:  # the baseline must never take the GPU number down with it
                    cands.append({"value": None, "unit": "grid-cell-timesteps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"})
            cands.sort(key=lambda c: -(c["value"] or 0.0))
            line["cpu_baseline"] = cands[0]
            line["cpu_baseline_other_engine"] = cands[1]
            if cands[0]["value"] and cands[1]["value"]:
                line["cpu_baseline"]["sample"] += (f" | the faster of two restatements of the reference's CPU engines ({cands[0]['value'] / cands[1]['value']:.1f}x the "
                                                   "other, see cpu_baseline_other_engine); a stated baseline, not the target")
        if world == 1 and not args.no_other_configs and (T, ny, nx) == (8760, 215, 1440):
            del cube
            torch.cuda.empty_cache()
            line["other_configs"] = run_other_configs(torch)
            # the same figures where the driver's record keeps them (it stores `roofline` whole): every BASELINE shape's fraction of
            # its bounding roofline, kernel time and whole-pass time
            line["roofline"]["other"] = {
                r["config"]: ({"bound": r["bound"], "frac": r["frac"], "hbm_frac": r.get("hbm_frac", r["frac"]), "kernel_ms_mean": r["kernel_ms_mean"],
                               "ms_per_step": r["ms_per_step"], "dtype": r["dtype"], "kernel": r["kernel"], "spatial_route": r.get("spatial_route")} if "error" not in r else {"error": r["error"]})
                for r in line["other_configs"]}
            if not args.no_ingest:
                try:
                    line["ingest"], line["end_to_end"] = run_store_jobs(torch, cpu=not args.no_cpu_baseline)
                    f0 = next(iter(line["ingest"]["fields"].values()))
                    line["roofline"]["ingest"] = {"workload": "configs[0] store (RAM) -> HBM, decoded GB/s, noisy bench field",
                                                  "hbm_decode_GBps": f0["chunks_decoded_in_hbm"]["GBps"],
                                                  "host_decode_GBps": f0["chunks_decoded_on_host_threads"]["GBps"]}
                    # SURVEY §8d (ii) where the driver's record keeps it: whole jobs from the store to the DataFrame, GPU (both decode routes) and CPU
                    line["roofline"]["end_to_end"] = {
                        "unit": "grid-cell-timesteps/s (store in RAM -> dataset_from_path -> aggregate_dataset -> DataFrame; T=8760 x 104x236 f32)",
                        **{("configs[0]" if nm.startswith("configs[0]") else "configs[1]"): {
                            "gpu_hbm_decode": j["chunks_decoded_in_hbm"]["value"], "gpu_host_decode": j["chunks_decoded_on_host_threads"]["value"],
                            "gpu_ms": j["chunks_decoded_in_hbm"]["total_ms"],
                            "cpu": (j.get("cpu") or {}).get("value"), "cpu_cores": (j.get("cpu") or {}).get("cores"), "cpu_ms": (j.get("cpu") or {}).get("total_ms")}
                           for nm, j in line["end_to_end"]["jobs"].items()}}
                except Exception as e:      # the ingest / end-to-end figures must never take the headline down with them
                    line["ingest"] = {"failed": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
