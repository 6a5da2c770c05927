"""Orchestration (`aggfly/cli/pipeline.py:124-172`): regions -> weights -> one
``aggregate_dataset`` per resolved ``{year}`` path -> concatenated panel -> writer.

The reference loops the years sequentially in one process.  Here the loop is the multi-GPU
time-shard scheduler: launched under ``torch.distributed.run`` each rank takes every
``world``-th path (a year is an outer-period-aligned time shard), reduces it on its own GPU,
and the per-year region x period panels are all-gathered (RCCL over xGMI) to rank 0.
A config with ONE un-templated store is cut along time inside the store instead (`run_store`):
ranks take runs of output periods and stream only their own chunks; a share beyond the HBM
budget (``AGGFLY_HIP_WINDOW_BYTES``, default 60 % of the free HBM) goes through in windows.
"""
from __future__ import annotations

import glob
import os

import numpy as np
import pandas as pd

import aggfly_amd as af
from aggfly_amd import distributed as D
from aggfly_amd.cfcalendar import CFTimeIndex
from aggfly_amd.weights import _read_table, georegions_from_path

from . import preprocess as preprocess_mod


def build_regions(config):
    return georegions_from_path(config.regions_path, config.regionid, config.region_list)


def _open_kwargs(config, georegions):
    """What every `dataset_from_path` call of a run shares."""
    kwargs = {}
    if config.chunks is not None:
        kwargs["chunks"] = config.chunks
    if config.storage_options is not None:
        kwargs["storage_options"] = config.storage_options
    if config.reader_engine is not None:
        kwargs["engine"] = config.reader_engine
    clip = georegions if config.clip_to_regions and _has_bounds(georegions) else None
    kwargs.update(xycoords=config.xycoords, timecoord=config.timecoord, georegions=clip, lon_is_360=config.lon_is_360,
                  preprocess=preprocess_mod.resolve_from_config(config), name=config.var)
    return kwargs


def store_route_ok(config, paths) -> bool:
    """One un-templated Zarr store / netCDF-4 file whose outputs share a nestable output frequency can be cut along
    time INSIDE the store: ranks (and, beyond the HBM budget, windows) take runs of output periods
    (`distributed.aggregate_store_sharded`)."""
    from .. import hip, io as afio
    if len(paths) != 1 or hip.device_count() == 0:
        return False
    p = paths[0]
    if not isinstance(p, str) or any(ch in p for ch in "*?[") or "://" in p:
        return False
    if not (afio._looks_like_zarr(p) or (os.path.isfile(p) and afio._is_hdf5(p))):
        return False
    try:
        D.output_freq(config.to_aggregator_dict())
    except ValueError:
        return False
    return True


def run_store(config, path, log=lambda m: None):
    """The single-store route: no sample load; every rank streams only its own output periods."""
    log(f"Loading regions: {config.regions_path}")
    georegions = build_regions(config)
    tpath = find_weights_table(config)
    log(f"Loading weights table: {tpath}")
    table = _read_table(tpath)

    def weights_of(ds):
        w = af.weights_from_objects(ds, georegions, table=table, project_dir=config.project_dir, zero_weight=config.zero_weight)
        w.calculate_weights()
        return w

    rank, world = D.world()
    if world > 1:
        # fewer output periods than ranks (one year, annual output): shard the cells instead — latitude bands, one all_reduce
        from aggfly_amd import io as afio
        from aggfly_amd.timegroups import resample_groups
        aggd = config.to_aggregator_dict()
        tindex = afio.read_time_coordinate(path, config.var, config.timecoord)
        if config.time_sel is not None:
            win = afio._time_window(tindex, config.time_sel)
            tindex = tindex[win[0]:win[1]] if win is not None else tindex
        P = len(resample_groups(tindex, D.output_freq(aggd))[1])
        if P < world:
            try:
                log(f"Aggregating {path}: {P} output period(s) < {world} ranks -> latitude bands, one all_reduce")
                return D.aggregate_store_cells(weights_of, path, config.var, aggd, time_sel=config.time_sel,
                                               **_open_kwargs(config, georegions))
            except ValueError as e:           # staged specs, mixed frequencies, fewer rows than ranks: time route
                log(f"cell sharding not applicable ({e}); falling back to time sharding")
    log(f"Aggregating {path}: output periods split over {world} rank(s), streamed through HBM in windows")
    budget = os.environ.get("AGGFLY_HIP_WINDOW_BYTES")
    try:
        return D.aggregate_store_sharded(weights_of, path, config.var, config.to_aggregator_dict(), engine=config.engine,
                                         max_window_bytes=int(budget) if budget else None, time_sel=config.time_sel,
                                         **_open_kwargs(config, georegions))
    except ValueError as e:
        if "time_sel" not in str(e):
            raise
        log(f"{e}; reading the store whole")
        return None


def load_dataset(config, path, georegions):
    kwargs = {}
    if config.chunks is not None:
        kwargs["chunks"] = config.chunks
    if config.storage_options is not None:
        kwargs["storage_options"] = config.storage_options
    if config.reader_engine is not None:
        kwargs["engine"] = config.reader_engine
    clip = georegions if config.clip_to_regions and _has_bounds(georegions) else None
    from .. import hip
    if hip.device_count() > 0 and "device" not in kwargs:
        # Zarr stores stream straight into HBM (native chunk decode, cached pinned staging); other
        # containers and codec chains fall back to the host route inside dataset_from_path
        kwargs["device"] = "cuda"
    return af.dataset_from_path(path, var=config.var, xycoords=config.xycoords, timecoord=config.timecoord,
                                time_sel=config.time_sel, georegions=clip, lon_is_360=config.lon_is_360,
                                preprocess=preprocess_mod.resolve_from_config(config), name=config.var, **kwargs)


def _has_bounds(georegions) -> bool:
    try:
        georegions.bounds
        return True
    except ValueError:
        return False


def find_weights_table(config) -> str:
    """The precomputed weights: ``weights.table`` or the single GridWeights ``.feather`` the
    reference cached under ``project_dir`` (`aggfly/cache/project_cache.py:46-47`)."""
    if config.weights_table:
        return config.weights_table
    if config.project_dir:
        hits = sorted(glob.glob(os.path.join(config.project_dir, "tmp", "GridWeights", "*", "*.feather")))
        if len(hits) == 1:
            return hits[0]
        if len(hits) > 1:
            raise FileNotFoundError(f"{len(hits)} cached weight tables under {config.project_dir}; set weights.table to pick one")
    raise FileNotFoundError(
        "no precomputed weights found: set weights.table, or point weights.project_dir at a cache written by "
        "aggfly's calculate_weights() (weights are computed on the CPU by aggfly.weights, not by this engine)")


def compute_weights(config, log=lambda m: None):
    log(f"Loading regions: {config.regions_path}")
    georegions = build_regions(config)
    path0 = config.resolved_paths()[0]
    log(f"Loading sample layer: {path0}")
    sample = load_dataset(config, path0, georegions)
    tpath = find_weights_table(config)
    log(f"Loading weights table: {tpath}")
    weights = af.weights_from_objects(sample, georegions, table=_read_table(tpath),
                                      project_dir=config.project_dir, zero_weight=config.zero_weight)
    weights.calculate_weights()
    return weights, georegions, sample


def run_pipeline(config, log=lambda m: None):
    """-> the panel DataFrame (on every rank when distributed)."""
    paths = config.resolved_paths()
    if store_route_ok(config, paths):
        df = run_store(config, paths[0], log)
        if df is not None:
            return df
    weights, georegions, sample = compute_weights(config, log)
    aggregator_dict = config.to_aggregator_dict()
    rank, world = D.world()
    mine = list(range(len(paths)))[rank::world]
    frames = {}
    for i in mine:
        log(f"Aggregating [{i + 1}/{len(paths)}] on rank {rank}: {paths[i]}")
        ds = sample if i == 0 else load_dataset(config, paths[i], georegions)
        frames[i] = af.aggregate_dataset(dataset=ds, weights=weights, aggregator_dict=aggregator_dict, engine=config.engine)
    if world > 1:
        frames = _gather_frames(frames, len(paths), config.regionid)
    ordered = [frames[i] for i in sorted(frames)]
    return pd.concat(ordered, ignore_index=True) if len(ordered) > 1 else ordered[0]


def _gather_frames(frames, n_paths, regionid):
    """All-gather the per-path panels as numeric tensors (values + time stamps + region row
    numbers); every rank rebuilds the full set of frames."""
    import torch
    import torch.distributed as dist
    rank, world = D.world()
    backend = dist.get_backend()
    dev = "cuda" if backend == "nccl" else "cpu"
    out = {}
    for i in range(n_paths):
        owner = i % world
        if owner == rank:
            df = frames[i]
            cols = [c for c in df.columns if c not in (regionid, "time")]
            t = df["time"]
            cal = getattr(t.iloc[0], "calendar", None) if len(t) else None
            if cal is None:
                tnum = pd.DatetimeIndex(t).asi8
            else:
                tnum = CFTimeIndex.from_fields([x.year for x in t], [x.month for x in t], [x.day for x in t],
                                               [x.hour for x in t], calendar=cal).seconds
            meta = [df[regionid].tolist(), cols, cal]
            # int64 time stamps ride along bit-for-bit, reinterpreted as float64
            vals = np.ascontiguousarray(np.column_stack(
                [np.asarray(tnum, dtype=np.int64).view(np.float64), df[cols].to_numpy(dtype=np.float64)])) \
                if len(df) else np.zeros((0, 1 + len(cols)))
            shape = torch.tensor([vals.shape[0], vals.shape[1]], dtype=torch.int64, device=dev)
        else:
            meta, shape = None, torch.zeros(2, dtype=torch.int64, device=dev)
        dist.broadcast(shape, src=owner)
        buf = torch.from_numpy(vals).to(dev) if owner == rank else torch.empty(tuple(shape.tolist()), dtype=torch.float64, device=dev)
        dist.broadcast(buf, src=owner)                      # the panel itself: RCCL on GPUs
        box = [meta]
        dist.broadcast_object_list(box, src=owner)          # region ids + column names: control plane
        ids, cols, cal = box[0]
        arr = buf.cpu().numpy()
        tn = np.ascontiguousarray(arr[:, 0]).view(np.int64)
        time = pd.DatetimeIndex(tn).values if cal is None else np.array(list(CFTimeIndex(tn, cal)), dtype=object)
        df = pd.DataFrame({regionid: ids, "time": time})
        for j, c in enumerate(cols):
            df[c] = arr[:, 1 + j]
        out[i] = df
    return out


def write_output(df, path, fmt):
    """`write_output` (`pipeline.py:159-172`)."""
    parent = os.path.dirname(path)
    if parent:
        os.makedirs(parent, exist_ok=True)
    if df["time"].dtype == object:        # CF calendars: ISO strings keep the stamp (e.g. Feb 30) portable
        df = df.assign(time=[str(t) for t in df["time"]])
    if fmt == "parquet":
        df.to_parquet(path, index=False)
    elif fmt == "feather":
        df.reset_index(drop=True).to_feather(path)
    elif fmt == "csv":
        df.to_csv(path, index=False)
    else:
        raise ValueError(f"unsupported output format: {fmt}")
