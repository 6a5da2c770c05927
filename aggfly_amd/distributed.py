"""Multi-GPU execution of the hot path: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU tests).

The reference has no distributed path of its own (dask only, `aggregate_utils.py:62`); what
shards here is the path itself:

* ``shard="time"`` — the time axis is cut at OUTER-period boundaries, so every (cell, period)
  window lives on exactly one GPU.  Each rank reduces its own periods to ``res[K, R, P_local]``
  and one ``all_gather`` assembles the region x period panel.  No halo, no all-to-all.
* ``shard="cells"`` — for specs with fewer output periods than GPUs (annual output of one
  year): each rank takes a latitude band; ``num`` and ``den`` are plain sums over cells, so one
  ``all_reduce(SUM)`` of ``(K+1)·R·P`` doubles finishes the job, then ``res = num / den``.

The exchange helpers (`gather_panel`, `reduce_num_den`) work on any backend and are covered by
world_size-2 gloo tests; only `aggregate_dataset_sharded` / `aggregate_store_sharded` / `aggregate_store_cells` touch the GPU.
`aggregate_store_sharded` is the north_star's "zarr chunks streamed per GPU": every rank opens the same
store, works out its own run of output periods from the time coordinate, and streams ONLY those time steps
into its GPU (`io.dataset_from_path(..., device=, time window)`) before the panel is gathered.
"""
from __future__ import annotations

import os

import numpy as np

from .timegroups import resample_groups


def _dist():
    import torch.distributed as dist
    return dist


def world(group=None):
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def choose_backend(n_devices: int, override=None) -> str:
    """"nccl" (RCCL over xGMI) when this NODE has one GPU per local rank, else "gloo" (a rehearsal of N ranks on fewer
    cards).  The local device count is compared with LOCAL_WORLD_SIZE — the ranks torchrun started on this node — not
    with the global WORLD_SIZE: 2 nodes x 8 GPUs (WORLD_SIZE 16) is an RCCL job.  ``override`` (an environment
    switch of the caller) wins."""
    if override:
        return override
    local = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE", "1"))
    return "nccl" if n_devices >= local else "gloo"


def init_local_rank(backend=None) -> dict:
    """One process per GPU (torchrun's RANK / LOCAL_RANK / WORLD_SIZE in the environment): select this rank's card —
    ``LOCAL_RANK`` modulo the visible devices, so only a rehearsal on fewer cards ever shares one — choose the backend
    against this NODE's device count (`choose_backend`: "nccl" = RCCL over xGMI when every local rank has its own GPU) and
    initialise the process group.  ``backend`` (e.g. from ``AGGFLY_DIST_BACKEND``) overrides the choice.  Does nothing but
    report when WORLD_SIZE is 1 or the group already exists.  -> {"backend", "device", "devices_visible", "world_size"}.

    Everything the engine allocates afterwards (cubes, weight tables, plans) follows the selected device; the handles are
    bound to it (include/aggfly_hip.h "Devices")."""
    import torch
    dist = _dist()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes needs dmabuf IPC on this pool's hosts
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % ndev if ndev else None
    if dev is not None:
        torch.cuda.set_device(dev)
    info = {"backend": None, "device": dev, "devices_visible": ndev, "world_size": ws}
    if dist.is_available() and dist.is_initialized():
        info["backend"] = dist.get_backend()
        return info
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        info["backend"] = choose_backend(ndev, backend)
        if info["backend"] == "nccl" and dev is not None:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(info["backend"])
    return info


def devices_used(group=None) -> dict:
    """Which cards do the ranks of the group really sit on?  -> {"ranks", "devices_used" (distinct (host, GPU uuid) pairs),
    "devices": [...]}: N ranks on RCCL must report N distinct devices; a rehearsal on one card reports 1."""
    import torch
    dist = _dist()
    rank, ws = world(group)
    if torch.cuda.is_available():
        d = torch.cuda.current_device()
        props = torch.cuda.get_device_properties(d)
        ident = (os.uname().nodename, str(getattr(props, "uuid", "")) or f"index{d}", props.name)
    else:
        ident = (os.uname().nodename, "cpu", "cpu")
    idents = [ident]
    if ws > 1:
        idents = [None] * ws
        dist.all_gather_object(idents, ident, group=group)
    return {"ranks": ws, "devices_used": len({i[:2] for i in idents}), "devices": [list(i) for i in idents]}


def split_even(n: int, rank: int, world_size: int):
    """Contiguous balanced split of range(n): -> (lo, hi)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def output_freq(aggregator_dict) -> str:
    """The output frequency of a spec (the last 'aggregate' step of every name must agree)."""
    from .engine import _agg_params
    freqs = set()
    for name, steps in aggregator_dict.items():
        last = None
        for kind, params in steps:
            if kind == "aggregate":
                last = _agg_params(params)[1]
        if last is None:
            raise ValueError(f"output {name!r} has no aggregate step")
        freqs.add(last)
        for kind, params in steps:
            if kind == "aggregate" and _agg_params(params)[1] == "W":
                raise ValueError("time sharding needs nested groupings (date/month/year); 'week' does not nest")
    if len(freqs) != 1:
        raise ValueError(f"all outputs must share one output frequency to shard the time axis, got {sorted(freqs)}")
    return freqs.pop()


def time_shard_bounds(tindex, freq: str, rank: int, world_size: int):
    """-> (k_lo, k_hi, p_lo, p_hi, P): this rank's time steps and output periods."""
    bounds, labels = resample_groups(tindex, freq)
    P = len(labels)
    p_lo, p_hi = split_even(P, rank, world_size)
    return int(bounds[p_lo]), int(bounds[p_hi]), p_lo, p_hi, P


def label_positions(local_labels, labels):
    """Position of every local output label in ``labels`` (exact match; a label that is not there is an error)."""
    import pandas as pd
    from .cfcalendar import CFTimeIndex
    if len(local_labels) == 0:
        return np.zeros(0, dtype=np.int64)
    if isinstance(labels, CFTimeIndex) or isinstance(local_labels, CFTimeIndex):
        if not (isinstance(labels, CFTimeIndex) and isinstance(local_labels, CFTimeIndex) and labels.calendar == local_labels.calendar):
            raise ValueError("local and global output labels are on different calendars")
        pos = np.searchsorted(labels.seconds, local_labels.seconds)
        ok = (pos < len(labels)) & (labels.seconds[np.minimum(pos, len(labels) - 1)] == local_labels.seconds)
    else:
        pos = pd.DatetimeIndex(labels).get_indexer(pd.DatetimeIndex(local_labels))
        ok = pos >= 0
    if not bool(np.all(ok)):
        raise ValueError("a shard produced an output period that the whole time axis does not have")
    return np.asarray(pos, dtype=np.int64)


def place_by_label(res_local, local_labels, labels, p_lo: int, p_hi: int):
    """``res_local[K, R, P_local]`` (periods ``local_labels``) -> a NaN-filled block ``[K, R, p_hi - p_lo]`` whose
    period axis is ``labels[p_lo:p_hi]``.

    A shard's own time slice runs from its first to its last time stamp, so resample bins of the whole axis that are
    EMPTY at the start or end of the share (seasonal data: June-August only; gaps on a shard or window boundary) do not
    exist locally: the local result has fewer periods than the share and must be placed by label, not by position.
    Periods without data stay NaN — what the unsharded run gives an empty resample bin (`nb_kernels.py:138-141`)."""
    import torch
    n = int(p_hi - p_lo)
    pos = label_positions(local_labels, labels) - int(p_lo)
    if len(pos) != res_local.shape[2]:
        raise ValueError(f"shard result has {res_local.shape[2]} periods but {len(pos)} labels")
    if len(pos) and (pos.min() < 0 or pos.max() >= n):
        raise ValueError("a shard produced an output period outside its own share of the time axis")
    if len(pos) == n and bool(np.array_equal(pos, np.arange(n))):
        return res_local
    block = torch.full((res_local.shape[0], res_local.shape[1], n), float("nan"), dtype=res_local.dtype, device=res_local.device)
    if len(pos):
        block.index_copy_(2, torch.as_tensor(pos, device=res_local.device), res_local)
    return block


def gather_panel(res_local, p_counts, group=None):
    """all_gather of per-rank ``res[K, R, P_local]`` tensors with different P_local (every rank's block must span
    exactly its share of the output periods: see `place_by_label`).

    Every rank pads to max(P_local), one all_gather moves the padded blocks (RCCL over xGMI
    on GPUs), and the blocks are trimmed and concatenated along the period axis.
    Returns the full ``res[K, R, P]`` on every rank."""
    import torch
    dist = _dist()
    rank, ws = world(group)
    if res_local.shape[2] != int(p_counts[rank]):
        raise ValueError(f"rank {rank}: local panel has {res_local.shape[2]} periods, its share has {int(p_counts[rank])}")
    if ws == 1:
        return res_local
    pmax = int(max(p_counts))
    K, R = res_local.shape[0], res_local.shape[1]
    pad = torch.full((K, R, pmax), float("nan"), dtype=res_local.dtype, device=res_local.device)
    pad[:, :, :res_local.shape[2]] = res_local
    blocks = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(blocks, pad.contiguous(), group=group)
    return torch.cat([b[:, :, :int(n)] for b, n in zip(blocks, p_counts)], dim=2)


def reduce_num_den(num, den, group=None):
    """Cell-axis sharding: sum the per-rank numerators/denominators, then divide
    (`aggfly/aggregate/spatial.py:127-133`).  -> (num, den, res) on every rank."""
    import torch
    dist = _dist()
    _, ws = world(group)
    if ws > 1:
        buf = torch.cat([num.reshape(-1), den.reshape(-1)])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        num = buf[:num.numel()].reshape(num.shape)
        den = buf[num.numel():].reshape(den.shape)
    if num.is_cuda:
        from . import hip
        res = hip.panel_divide(num, den)              # the library's divide (afhip_panel_divide)
    else:                                             # host tensors: the gloo rehearsal of the exchange in the CPU tests
        res = torch.where(den.unsqueeze(0) != 0, num / den.unsqueeze(0), torch.full_like(num, float("nan")))
    return num, den, res


def band_csr_triplets(rows, cols, w, ny, nx, y0, y1):
    """Restrict COO triplets (cols on the stored ny x nx grid) to the latitude band [y0, y1) and
    re-index the columns onto the band."""
    iy = cols // nx
    keep = (iy >= y0) & (iy < y1)
    return rows[keep], cols[keep] - y0 * nx, w[keep]


def _cells_band_panel(weights, grid_ds, cube_band, y0, y1, tindex, aggregator_dict, group=None):
    """One rank's latitude band ``[y0, y1)`` of a cell-sharded job: the band's cube (time, y1 - y0, nx) is reduced
    against the weights restricted to the band, and ONE all_reduce(SUM) of the numerators and denominators
    finishes the panel (`reduce_num_den`).  ``grid_ds`` carries the full grid (its longitude order is folded into
    the CSR columns as in `engine.get_csr`).  -> (res[K, R, P] on every rank, names, region_ids, labels)"""
    from . import aggregate as agg, engine as eng, hip
    ny, nx = len(grid_ds.latitude), len(grid_ds.longitude)
    _, fused_cols, staged, names = agg._lower_all(aggregator_dict)
    if staged:
        raise ValueError("cell sharding supports fused (two-level) specs only")
    groups = eng.plan_groups(tindex, fused_cols)
    if len(groups) != 1:
        raise ValueError("cell sharding needs all outputs to share their group frequencies")
    cols, ib, ob, labels = groups[0]
    wrows, wcols, wv, region_ids = eng.weight_triplets(weights.weights, np.asarray(weights.grid.cell_id))
    lon_order, _ = grid_ds.lon_order_to_180()
    iy, ixs = np.divmod(wcols, nx)
    wcols_mem = iy * nx + lon_order[ixs]
    br, bc, bw = band_csr_triplets(wrows, wcols_mem, wv, ny, nx, y0, y1)
    csr = hip.CSR(br, bc, bw, len(region_ids), (y1 - y0) * nx, device=cube_band.device)
    pr = eng.run_fused_pass(cube_band, cols, ib, ob, csr=csr, want_cells=False)
    if len(pr) != 1:
        raise hip.HipUnsupported("cell sharding needs the spec to fit one fused pass")
    _, _, full = reduce_num_den(pr[0].panel["num"], pr[0].panel["den"], group)
    return full, names, region_ids, labels


def aggregate_dataset_sharded(weights, dataset=None, aggregator_dict=None, engine="auto", shard="time",
                              group=None, **kwargs):
    """`aggregate_dataset` across the GPUs of the process group.  Every rank passes the same
    dataset description; each reduces its shard; the assembled frame is returned on every rank.
    """
    import torch
    from . import aggregate as agg, engine as eng, hip
    if dataset is None:
        raise ValueError("No dataset provided.")
    if aggregator_dict is None:
        aggregator_dict = kwargs
    rank, ws = world(group)
    tindex = dataset.da.coords["time"]
    freq = output_freq(aggregator_dict)
    if shard == "time":
        k_lo, k_hi, p_lo, p_hi, P = time_shard_bounds(tindex, freq, rank, ws)
        counts = [split_even(P, r, ws)[1] - split_even(P, r, ws)[0] for r in range(ws)]
        _, labels = resample_groups(tindex, freq)
        local = dataset.deepcopy()
        local.da = dataset.da.isel(time=slice(k_lo, k_hi))
        _, fused_cols, _, names = agg._lower_all(aggregator_dict)
        csr, region_ids = eng.get_csr(weights, dataset, device=eng.dataset_device(dataset))
        if k_hi > k_lo:
            res, names, region_ids, local_labels = agg.panel_arrays(weights, local, aggregator_dict, engine)
        else:
            res, local_labels = torch.empty((len(names), len(region_ids), 0), dtype=torch.float64, device="cuda"), labels[:0]
        full = gather_panel(place_by_label(res, local_labels, labels, p_lo, p_hi), counts, group)
    elif shard == "cells":
        ny = len(dataset.latitude)
        y0, y1 = split_even(ny, rank, ws)
        cube = eng.device_cube(dataset)[:, y0:y1, :].contiguous()
        full, names, region_ids, labels = _cells_band_panel(weights, dataset, cube, y0, y1, tindex, aggregator_dict, group)
    else:
        raise ValueError("shard must be 'time' or 'cells'")
    df = agg._assemble_frame(full, names, region_ids, labels, weights)
    return agg._merge_regions(df, weights)


def _step_bytes(path, var):
    """Bytes one time step of ``var`` takes in HBM (full stored grid: a clipped read takes less), or None when
    the container's shape cannot be told without reading it."""
    import numpy as np
    from . import io as afio
    try:
        if afio._looks_like_zarr(path):
            za = afio.ZarrArray(os.path.join(path, var))
            shape, dt = za.shape, np.dtype(za.dtype)
        elif afio._is_hdf5(path):
            from . import hdf5
            with hdf5.H5File(path) as f:
                shape, dt = f.datasets[var].shape, np.dtype(f.datasets[var].dtype)
        else:
            return None
        item = dt.itemsize if dt.kind == "f" else 4           # packed integers are unpacked to float32 in HBM
        return int(np.prod(shape[1:], dtype=np.int64)) * item
    except Exception:
        return None


def plan_windows(bounds, p_lo, p_hi, step_bytes, budget_bytes):
    """Cut the output periods ``[p_lo, p_hi)`` into consecutive runs whose time steps fit ``budget_bytes`` in HBM
    (a run always holds at least one period).  -> [(q_lo, q_hi), ...]"""
    if p_hi <= p_lo:
        return []
    if not step_bytes or not budget_bytes:
        return [(p_lo, p_hi)]
    runs, q = [], p_lo
    while q < p_hi:
        e = q + 1
        while e < p_hi and (int(bounds[e + 1]) - int(bounds[q])) * step_bytes <= budget_bytes:
            e += 1
        runs.append((q, e))
        q = e
    return runs


def aggregate_store_cells(weights_of, path, var, aggregator_dict, group=None, **open_kwargs):
    """Cell-sharded `aggregate_dataset` straight from a store: for jobs with fewer output periods than GPUs (one
    annual period: BASELINE configs[0], [1], [4]).  Every rank opens the store's coordinates (no data), takes a band of
    latitude rows, streams ONLY the chunks that touch its band into HBM, reduces it against the weights of the band,
    and one all_reduce(SUM) of ``(K+1) x R x P`` doubles assembles the panel on every rank."""
    from . import aggregate as agg, engine as eng, io as afio
    rank, ws = world(group)
    head_kwargs = {k: v for k, v in open_kwargs.items() if k != "time_sel"}                            # (no steps to select from)
    head = afio.dataset_from_path(path, var, device="cuda", time_window=(0, 0), **head_kwargs)       # grid and coordinates only
    weights = weights_of(head)
    if len(head.latitude) < ws:
        raise ValueError(f"{ws} ranks for {len(head.latitude)} latitude rows: use fewer ranks or time sharding")
    y0, y1 = split_even(len(head.latitude), rank, ws)
    band = afio.dataset_from_path(path, var, device="cuda", lat_window=(y0, y1), **open_kwargs)
    if len(band.latitude) != y1 - y0 or len(band.longitude) != len(head.longitude):
        raise RuntimeError("the band read does not match the grid the weights were laid on")
    tindex = band.da.coords["time"]
    full, names, region_ids, labels = _cells_band_panel(weights, head, eng.device_cube(band), y0, y1, tindex, aggregator_dict, group)
    df = agg._assemble_frame(full, names, region_ids, labels, weights)
    return agg._merge_regions(df, weights)


def aggregate_store_sharded(weights_of, path, var, aggregator_dict, engine="auto", group=None, max_window_bytes=None,
                            **open_kwargs):
    """Time-sharded `aggregate_dataset` straight from a store on disk.

    ``weights_of(dataset) -> GridWeights`` builds the weights for the rank's (possibly region-clipped) dataset
    (e.g. ``lambda ds: af.weights_from_objects(ds, regions, table=table)``).  Every rank reads the store's
    time coordinate, takes the output periods `split_even` gives it, streams only those steps into HBM, reduces
    them, and one all_gather assembles the region x period panel; the frame is returned on every rank.
    A rank's share that does not fit in HBM is taken in consecutive windows of whole output periods
    (``max_window_bytes``; default: 60 % of the free HBM), each streamed, reduced and released before the next:
    a store of any length runs on one GPU.
    ``open_kwargs`` go to `dataset_from_path` (``xycoords``, ``lon_is_360``, ``georegions``, ``preprocess`` ...);
    ``time_sel`` among them restricts the job to the selected years before the periods are dealt out."""
    import torch
    from . import aggregate as agg, io as afio
    rank, ws = world(group)
    timecoord = open_kwargs.get("timecoord", "time")
    tindex = afio.read_time_coordinate(path, var, timecoord)
    # a time selection (`Dataset(time_sel=...)`, `dataset.py:88-92`) narrows the job to one contiguous run of steps first
    k_off, time_sel = 0, open_kwargs.pop("time_sel", None)
    if time_sel is not None:
        win = afio._time_window(tindex, time_sel)
        if win is None:
            raise ValueError("time_sel does not pick one contiguous run of this store's time axis")
        k_off, tindex = int(win[0]), tindex[int(win[0]):int(win[1])]
    freq = output_freq(aggregator_dict)
    bounds, labels = resample_groups(tindex, freq)
    P = len(labels)
    counts = [split_even(P, r, ws)[1] - split_even(P, r, ws)[0] for r in range(ws)]
    p_lo, p_hi = split_even(P, rank, ws)
    names = agg._lower_all(aggregator_dict)[3]
    if max_window_bytes is None and torch.cuda.is_available():
        max_window_bytes = int(0.6 * torch.cuda.mem_get_info()[0])
    windows = plan_windows(bounds, p_lo, p_hi, _step_bytes(path, var), max_window_bytes)
    weights, parts, region_ids = None, [], None
    # an empty share still opens the store (coordinates, grid) so that every rank builds the same weights
    for q_lo, q_hi in (windows or [(p_lo, p_lo)]):
        k_lo, k_hi = k_off + int(bounds[q_lo]), k_off + int(bounds[q_hi])
        local = afio.dataset_from_path(path, var, device="cuda", time_window=(k_lo, k_hi), **open_kwargs)
        if weights is None:
            weights = weights_of(local)
        if k_hi > k_lo:
            res, names, region_ids, local_labels = agg.panel_arrays(weights, local, aggregator_dict, engine)
        else:
            from . import engine as eng
            _, region_ids = eng.get_csr(weights, local, device=eng.dataset_device(local))
            res, local_labels = torch.empty((len(names), len(region_ids), 0), dtype=torch.float64, device="cuda"), labels[:0]
        # by label: a window that starts or ends on empty resample bins has fewer local periods than q_hi - q_lo
        parts.append(place_by_label(res, local_labels, labels, q_lo, q_hi))
        # the window's kernels must have finished reading the cube before its block goes back to the allocator:
        # the next window's copy stream writes into (very likely) the same block
        torch.cuda.current_stream().synchronize()
        del local                                                      # the window's cube is released before the next is read
    res = parts[0] if len(parts) == 1 else torch.cat(parts, dim=2)
    full = gather_panel(res, counts, group)
    df = agg._assemble_frame(full, names, region_ids, labels, weights)
    return agg._merge_regions(df, weights)
