#!/usr/bin/env python3
"""Headline benchmark: grid-cell-timesteps/s of the fused transform + weighted reduce.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], shapes per SURVEY.md §8d): one year of hourly ERA5-like
2 m temperature on the 0.25 deg US-counties extent (T = 8760, 215 x 1440 = 309,600 cells,
fp64, synthetic), ~3.1k regions with area weights, and the fused plan

    dd[10,30]@date -> sum@year      +      mean@date -> power[1..4] -> sum@year      (K = 5)

A "step" is one whole pass of the hot path over one resident year: fused temporal kernel,
slot merge + shared validity, CSR weighted sums, divide (and, for N > 1, the RCCL
all-gather of the region x period panel, which runs on the collective's stream beside the
next step's kernels; every gather completes inside the timed region).  Inputs are resident in HBM before the timed
region.  With N > 1 every rank owns a different year (time-axis sharding on outer-period
boundaries, weak scaling): value = N * T * cells / max-over-ranks time.

The JSON line also carries
  roofline      the fused temporal kernel against the HBM peak: algorithmic bytes
                (T * cells * 8 B per launch) / the kernel's mean duration,
                measured with HIP events around every launch of the timed region;
  cpu_baseline  the plain-C port of the reference's numba engine (oracle/c) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes: this pool's driver only supports dmabuf IPC

HBM_PEAK_GBPS = 8000.0       # MI355X spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy rate


def c2_columns():
    cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    cols += [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    return cols


def make_cube(torch, T, ny, nx, dtype, seed):
    """ERA5-like synthetic temperatures generated on the device, slab by slab."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    cube = torch.empty((T, ny, nx), dtype=dtype, device="cuda")
    lat = torch.linspace(0.6, 1.4, ny, device="cuda", dtype=torch.float64)[None, :, None]
    for k0 in range(0, T, 256):
        k1 = min(T, k0 + 256)
        k = torch.arange(k0, k1, device="cuda", dtype=torch.float64)
        base = 15.0 + 12.0 * torch.sin(2 * np.pi * torch.floor(k / 24) / 365.0) + 6.0 * torch.sin(2 * np.pi * (k % 24) / 24.0)
        noise = torch.randn((k1 - k0, ny, nx), generator=g, device="cuda", dtype=torch.float32).to(torch.float64) * 3.0
        cube[k0:k1] = (base[:, None, None] * lat + noise).to(dtype)
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
                      f"(C/OpenMP port of the reference's numba engine, oracle/c)"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ny", type=int, default=215)
    ap.add_argument("--nx", type=int, default=1440)
    ap.add_argument("--T", type=int, default=8760)
    ap.add_argument("--regions", type=int, default=3100)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from aggfly_amd import hip, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)          # rehearsal of N > 1 on a 1-GPU box shares the card
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL needs one GPU per rank; AGGFLY_BENCH_BACKEND=gloo rehearses the N > 1 code path on one card
        backend = os.environ.get("AGGFLY_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    hip.require_gpu()

    T, ny, nx = args.T, args.ny, args.nx
    C = ny * nx
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    elem = 8 if args.dtype == "f64" else 4
    cube = make_cube(torch, T, ny, nx, dtype, seed=20260101 + rank)          # rank r owns year r
    ib = synth.hourly_bounds(T)
    ob = np.array([0, len(ib) - 1], dtype=np.int64)
    tab = synth.weights_table(ny, nx, args.regions, seed=7)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, C)
    cols = c2_columns()
    K = len(cols)
    plan = hip.FusedPlan(T, C, hip.F64 if args.dtype == "f64" else hip.F32, ib, ob, cols)
    out = {"num": torch.empty((K, R, 1), dtype=torch.float64, device="cuda"),
           "den": torch.empty((R, 1), dtype=torch.float64, device="cuda"),
           "res": torch.empty((K, R, 1), dtype=torch.float64, device="cuda")}
    # N > 1: the region x period panel of every step is all-gathered (RCCL over xGMI).  Two result buffers, so that
    # the gather of step i (on the collective's own stream) overlaps the kernels of step i + 1; a buffer is reused
    # only after its gather has finished, and every gather is waited for inside the timed region.
    pipelined = world > 1 and os.environ.get("AGGFLY_BENCH_PIPELINE", "1") != "0"
    outs = [out] + ([{k: torch.empty_like(v) for k, v in out.items()}] if pipelined else [])
    use_flat = world > 1 and dist.get_backend() == "nccl"
    if use_flat:        # one RCCL call straight into [world, K, R, 1]
        gathered = [torch.empty((world,) + tuple(out["res"].shape), dtype=torch.float64, device="cuda") for _ in outs]
    else:
        gathered = [[torch.empty_like(out["res"]) for _ in range(world)] for _ in outs] if world > 1 else None
    pending = [None] * len(outs)

    def step(i):
        b = i % len(outs)
        if pending[b] is not None:
            pending[b].wait()                                 # the gather that last read this buffer
            pending[b] = None
        plan.run(cube, csr, out=outs[b])
        if world > 1:
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
    if world > 1:       # outside the timed region: the gathered panel really holds this rank's result of the last step
        b = (args.steps - 1) % len(outs)
        mine = gathered[b][rank]
        if not torch.equal(torch.nan_to_num(mine), torch.nan_to_num(outs[b]["res"])):
            raise SystemExit(f"rank {rank}: gathered panel differs from the local result")
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        traffic = None
        try:   # PMC-derived HBM bytes per launch for this workload, collected in a separate rocprofv3 --pmc pass
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = tj.get(f"c2_{args.dtype}_T{T}_C{C}", {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        ms_per_step = dt / args.steps * 1e3
        value = world * T * C * args.steps / dt
        k_ms = float(np.mean(kms)) if kms else float("nan")
        achieved = T * C * elem / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "grid-cell-timesteps/s (fused transform+weighted reduce)",
            "value": value, "unit": "grid-cell-timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ERA5-like hourly t2m 0.25deg, 1 year per GPU, US-counties extent "
                                   f"{ny}x{nx} cells, {R} regions area weights, fused dd[10,30]@date->sum@year + "
                                   "mean@date->power[1..4]->sum@year, %s, K=5" % ("fp64" if args.dtype == "f64" else "fp32 storage / fp64 accumulation"),
                       "T": T, "cells": C, "regions": R, "columns": K, "nnz": int(csr.nnz),
                       "sharding": ("time axis, one year per GPU; RCCL all-gather of the panel"
                                    + (", overlapped with the next step's kernels" if pipelined else "")) if world > 1 else "single GPU",
                       "plan": plan.describe()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_fused_temporal", "kernel_ms_mean": k_ms, "launches": len(kms),
                         "algorithmic_bytes_per_launch": T * C * elem},
        }
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
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
