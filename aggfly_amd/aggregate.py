"""Public aggregation API — same names, arguments and error behaviour as the reference's
`aggfly/aggregate/aggregate.py`, `temporal.py` and `spatial.py`, executed by the HIP engine.

    aggregate_dataset(weights, dataset=None, aggregator_dict=None, dataset_dict=None,
                      engine="auto", **kwargs) -> pd.DataFrame        aggregate.py:210-282
    aggregate_time(dataset, weights=None, aggregator_dict=None, engine="auto", **kwargs)
                                                                       aggregate.py:101-162
    aggregate_space(dataset_dict, weights, npartitions=None, **kwargs) aggregate.py:165-198
    TemporalAggregator(calc, groupby, ddargs=None, pre_compute=False, engine="auto")
                                                                       temporal.py:19-263
    SpatialAggregator(dataset, weights, names).compute()               spatial.py:37-154

``engine`` accepts the reference's names plus ``"hip"``.  Every valid name runs the HIP
engine (there is no dask or numba here); an unknown name raises ``ValueError`` like
`resolve_engine` (`aggfly/aggregate/nb_kernels.py:59-74`).
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Union

import numpy as np
import pandas as pd

from . import engine as eng
from . import hip
from .cfcalendar import CFTimeIndex
from .dataarray import DataArray
from .dataset import Dataset
from .timegroups import resample_groups, translate_groupby

ENGINES = ("auto", "hip", "dask", "numba")
_DEPRECATED_CLUSTER_KWARGS = ("n_workers", "threads_per_worker", "processes", "memory_limit", "cluster_args")
_warned_cpu_engine = False


def resolve_engine(engine: str, da=None, calc: str = "mean") -> str:
    """Engine selector seam (`nb_kernels.py:59-74`).  All valid names resolve to "hip"."""
    global _warned_cpu_engine
    if engine not in ENGINES:
        raise ValueError(f"engine must be 'dask', 'numba', 'hip' or 'auto', got {engine!r}")
    if engine in ("dask", "numba") and not _warned_cpu_engine:
        _warned_cpu_engine = True
        warnings.warn(f"engine={engine!r} is kept for API compatibility; this build has no CPU engines "
                      "and runs the HIP (MI355X) engine", RuntimeWarning, stacklevel=3)
    return "hip"


# dask-client helpers of the reference (`aggfly/aggregate/aggregate_utils.py:38-102`): there is
# no dask scheduler to manage; kept so existing scripts and the CLI import cleanly.
def distributed_client():
    return None


def is_distributed() -> bool:
    return False


def start_dask_client(n_workers: int = 2, threads_per_worker: int = 2, cap_numba_threads: int = 1, **kwargs):
    warnings.warn("start_dask_client is a no-op: the HIP engine does not use dask", RuntimeWarning, stacklevel=2)
    return None


def shutdown_dask_client():
    return None


# --------------------------------------------------------------------------------------
# temporal
# --------------------------------------------------------------------------------------
def _labels_of(ds: Dataset):
    return ds.da.coords["time"]


def _time_major(ds: Dataset):
    return eng.device_cube(ds)


def _dataset_from_cells(template: Dataset, cells_pc, labels, history) -> Dataset:
    """Wrap a [P, n_cells] device tensor as a Dataset on the template's grid."""
    ny, nx = len(template.latitude), len(template.longitude)
    data = cells_pc.reshape(cells_pc.shape[0], ny, nx)
    if eng.config.match_reference_f32 and str(template.da.dtype).endswith("float32"):
        import torch
        data = data.to(torch.float32)      # values are already float32-rounded: the cast is exact
    da = DataArray(data, ("time", "latitude", "longitude"),
                   {"time": labels, "latitude": template.latitude, "longitude": template.longitude})
    new = template.deepcopy()
    new.da = da
    new.history = list(history)
    return new


class TemporalAggregator:
    """One grouped temporal reduction (`temporal.py:19-263`)."""

    def __init__(self, calc: str, groupby: str, ddargs=None, pre_compute: bool = False, engine: str = "auto"):
        if calc not in eng.HIP_CALCS:
            raise ValueError(f"unknown calc {calc!r}; supported: {sorted(eng.HIP_CALCS)}")
        self.calc = calc
        self.groupby = translate_groupby(groupby)          # KeyError on unknown names, temporal.py:456
        self.ddargs = ddargs
        self.multi_dd = ddargs is not None and np.array(ddargs).ndim > 1
        self.pre_compute = pre_compute
        self.engine = engine
        self.kwargs = {"ddargs": ddargs} if calc in eng.THR_CALCS else {}

    def execute(self, dataset: Dataset, weights=None, update: bool = False, **kwargs):
        """Run this step on ``dataset`` (single level) and return a Dataset, or a list of
        Datasets for a multi-row ``ddargs`` (`temporal.py:165-263`)."""
        resolve_engine(self.engine, None, self.calc)
        tindex = _labels_of(dataset)
        if self.groupby == "W" and isinstance(tindex, CFTimeIndex):                # temporal.py:221-227
            raise NotImplementedError(
                "groupby='week' is not supported on non-standard CF calendars (noleap/360_day/etc.): "
                "xarray/cftime has no weekly offset. Use 'date', 'month', or 'year'.")
        if len(dataset.da.dims) != 3:
            raise ValueError(f"the temporal engine expects 2 spatial dims, got {dataset.da.dims}")
        cube = _time_major(dataset)
        bounds, labels = resample_groups(tindex, self.groupby)
        rows = np.atleast_2d(np.asarray(self.ddargs, dtype=float)) if self.ddargs is not None else [None]
        cols = [eng.ColumnProg("x", eng.AggStep(self.calc, self.groupby, None if r is None else tuple(r))) for r in rows]
        ob = np.arange(len(bounds), dtype=np.int64)
        outs = []
        for i in range(0, len(cols), eng.MAX_COLS_PER_PASS):
            for pr in eng.run_fused_pass(cube, cols[i:i + eng.MAX_COLS_PER_PASS], bounds, ob):
                outs.extend(pr.cells[j] for j in range(pr.cells.shape[0]))
        hist = list(dataset.history) + [self.groupby]
        res = [_dataset_from_cells(dataset, c, labels, hist) for c in outs]
        if self.multi_dd:
            return res[0] if len(res) == 1 else res
        if update:
            dataset.da, dataset.history = res[0].da, res[0].history
            return dataset
        return res[0]


def _apply_transform(ds: Dataset, key: str, params: dict):
    """`transform_dataset` (`aggregate.py:36-78`) on a device-resident Dataset (staged path)."""
    if "exp" in params:
        exp = params["exp"]
        if not isinstance(exp, list):
            exp = [exp]
        return [ds.power(e) for e in exp[0]], [f"{key}_{e}" for e in exp[0]]
    if "inter" in params:
        return [ds.interact(params["inter"])], [key]
    if "spline" in params.get("transform", ""):
        return list(ds.spline()), [f"{key}_spline{x}" for x in (1, 2)]
    raise ValueError("No valid transform argument provided.")


def _staged_name(dataset: Dataset, key: str, steps, engine: str) -> Dict[str, Dataset]:
    """The reference's interpreter loop, one GPU step at a time (`aggregate.py:131-158`)."""
    keys, data = [key], [dataset.deepcopy().to_device()]     # every step below runs in HBM
    for kind, params in steps:
        if kind == "aggregate":
            agg = params if isinstance(params, TemporalAggregator) else TemporalAggregator(**params, engine=engine)
            data = [agg.execute(x) for x in data]
            if agg.multi_dd:
                if len(data) > 1:
                    raise ValueError("Cannot aggregate multiple datasets with multiple ddargs, "
                                     "e.g., multiple polynomials for multiple bins")
                d0 = data[0]
                data = d0 if isinstance(d0, list) else [d0]
                keys = [f"{key}_{x[0]}_{x[1]}" for x in agg.ddargs]
        elif kind == "transform":
            nd, nk = [], []
            for d, k in zip(data, keys):
                d2, k2 = _apply_transform(d, k, params)
                nd.extend(d2)
                nk.extend(k2)
            data, keys = nd, nk
        else:
            raise ValueError(f"unknown step type {kind!r}")
    return dict(zip(keys, data))


def _lower_all(aggregator_dict):
    """-> (order, fused ColumnProgs, names that must run staged).

    Output keys follow dict semantics like the reference (`aggregate.py:160-161`:
    ``out_dict = out_dict | dict(zip(keys, data))``): when two columns produce the same key —
    e.g. two ``ddargs`` rows with equal bounds but different flags — the LATER column wins and
    keeps the FIRST one's position; the shadowed column is never computed and takes no part in
    the shared validity mask."""
    lowered, staged = [], []
    for name, steps in aggregator_dict.items():
        cols, fusable = eng.lower_spec(name, steps)
        lowered.append((name, cols, fusable))
        if not fusable:
            staged.append(name)
    winner = {}                                   # key -> (name, index in that name's column list)
    for name, cols, _ in lowered:
        for i, c in enumerate(cols):
            winner[c.key] = (name, i)             # later overrides, first position kept by dict order
    fused_cols = []
    for name, cols, fusable in lowered:
        if fusable:
            fused_cols.extend(c for i, c in enumerate(cols) if winner[c.key] == (name, i))
    order = [(name, [k for k, (n, _) in winner.items() if n == name], fusable) for name, _, fusable in lowered]
    # keys in global first-insertion order
    order_keys = list(winner)
    return order, fused_cols, staged, order_keys


def transform_dataset(dataset: Dataset, key: str, **kwargs):
    """`transform_dataset` (`aggregate.py:36-78`), the eager form: ``(datasets, keys)`` for one 'transform' step.
    ``exp`` fans out to ``key_{e}`` over ``exp[0]`` (the reference indexes the wrapped list), ``inter`` keeps the
    key, ``transform='spline'`` gives ``key_spline1`` / ``key_spline2``.  `aggregate_dataset` does not call this —
    `engine.lower_spec` lowers the same steps into the fused plan — it is here for callers that use it directly."""
    if "exp" in kwargs:
        exps = kwargs["exp"] if isinstance(kwargs["exp"], list) else [kwargs["exp"]]
        out = {f"{key}_{e}": dataset.power(e) for e in exps[0]}
    elif "inter" in kwargs:
        out = {key: dataset.interact(kwargs["inter"])}
    elif "spline" in kwargs.get("transform", ""):
        out = dict(zip([f"{key}_spline1", f"{key}_spline2"], dataset.spline()))
    else:
        raise ValueError("No valid transform argument provided.")
    return out.values(), out.keys()


def multi_dd_to_dict(data, key, ddargs):
    """`multi_dd_to_dict` (`aggregate.py:285-303`): ``(data, [key_{lo}_{hi} ...])`` for a multi-threshold step."""
    return data, [f"{key}_{x[0]}_{x[1]}" for x in ddargs]


def aggregate_time(dataset: Dataset, weights=None, aggregator_dict=None, engine: str = "auto", **kwargs) -> Dict[str, Dataset]:
    """`aggregate_time` (`aggregate.py:101-162`): {output name: Dataset on the output time axis}."""
    resolve_engine(engine)
    if aggregator_dict is None:
        if kwargs is None:
            raise ValueError("No arguments provided.")
        aggregator_dict = kwargs
    tindex = _labels_of(dataset)
    order, fused_cols, staged, order_keys = _lower_all(aggregator_dict)
    _guard_week(fused_cols, tindex)
    results: Dict[str, Dataset] = {}
    if fused_cols:
        cube = _time_major(dataset)
        for cols, ib, ob, labels in eng.plan_groups(tindex, fused_cols):
            for pr in eng.run_fused_pass(cube, cols, ib, ob):
                for j, k in enumerate(pr.keys):
                    results[k] = _dataset_from_cells(dataset, pr.cells[j], labels, dataset.history)
    owner = {k: name for name, keys, _ in order for k in keys}
    for name in staged:
        for k, v in _staged_name(dataset, name, aggregator_dict[name], engine).items():
            if owner.get(k) == name:
                results[k] = v
    return {k: results[k] for k in order_keys}


def _guard_week(cols, tindex):
    if isinstance(tindex, CFTimeIndex):
        for c in cols:
            if c.inner.freq == "W" or (c.outer is not None and c.outer.freq == "W"):
                raise NotImplementedError(
                    "groupby='week' is not supported on non-standard CF calendars (noleap/360_day/etc.): "
                    "xarray/cftime has no weekly offset. Use 'date', 'month', or 'year'.")


# --------------------------------------------------------------------------------------
# spatial
# --------------------------------------------------------------------------------------
def _same_labels(a, b) -> bool:
    if isinstance(a, CFTimeIndex) or isinstance(b, CFTimeIndex):
        return isinstance(a, CFTimeIndex) and isinstance(b, CFTimeIndex) and a == b
    return len(a) == len(b) and bool(np.all(np.asarray(a) == np.asarray(b)))


def union_labels(axes):
    """The output time axis of several names: the sorted UNION of their labels, as the reference's
    `xr.combine_by_coords` (an outer join, `aggfly/aggregate/spatial.py:90-92`) builds it, and each axis' positions in it."""
    if all(isinstance(t, CFTimeIndex) for t in axes):
        if len({t.calendar for t in axes}) != 1:
            raise ValueError("output variables are on different calendars")
        secs = np.unique(np.concatenate([t.seconds for t in axes]))
        return CFTimeIndex(secs, axes[0].calendar), [np.searchsorted(secs, t.seconds) for t in axes]
    if any(isinstance(t, CFTimeIndex) for t in axes):
        raise ValueError("output variables mix a CF calendar with a standard one")
    idx = [pd.DatetimeIndex(t) for t in axes]
    u = idx[0]
    for t in idx[1:]:
        u = u.union(t)
    return u, [u.get_indexer(t) for t in idx]


def _aligned_cells(datasets):
    """-> (x [K, n_cells, P] float64 in HBM, labels): every name's [P_k, ny, nx] result laid on the union of the names'
    output labels.  Where a name has no value for a label the reference's outer join fills NaN (`spatial.py:90-97`); the
    shared validity mask then voids that period for EVERY name (`spatial.py:114-119`) and its rows are dropped — only
    placement happens here (index copies), the masking is the spatial kernels'."""
    import torch
    axes = [_labels_of(d) for d in datasets]
    same = all(_same_labels(axes[0], a) for a in axes[1:])
    labels, pos = (axes[0], None) if same else union_labels(axes)
    xs = []
    for i, d in enumerate(datasets):
        c = eng.device_cube(d)                                   # [P_k, ny, nx]
        c = c.reshape(c.shape[0], -1).to(torch.float64)
        if not same:
            full = torch.full((len(labels), c.shape[1]), float("nan"), dtype=torch.float64, device=c.device)
            full.index_copy_(0, torch.as_tensor(np.asarray(pos[i], dtype=np.int64), device=c.device), c)
            c = full
        xs.append(c.t())
    return torch.stack(xs).contiguous(), labels


def _label_values(labels):
    if isinstance(labels, CFTimeIndex):
        arr = np.empty(len(labels), dtype=object)
        for i, t in enumerate(labels):
            arr[i] = t
        return arr
    return pd.DatetimeIndex(labels).values


_ZERO_REGIONS: dict = {}


def _zero_weight_regions(wdf) -> set:
    """Regions whose weights sum to zero (`spatial.py:148-150`), cached per table object."""
    key = id(wdf)
    hit = _ZERO_REGIONS.get(key)
    if hit is not None and hit[0] == len(wdf):
        return hit[1]
    wsum = wdf.groupby("index_right")["weight"].sum()
    zr = set(wsum.index[~(wsum > 0)])
    try:
        import weakref
        weakref.finalize(wdf, _ZERO_REGIONS.pop, key, None)
        _ZERO_REGIONS[key] = (len(wdf), zr)
    except TypeError:
        pass
    return zr


def _download(t, wait=True):
    """HBM tensor -> numpy (``wait=False``: the copy is queued on the current stream; the caller synchronises before reading).  Large results go through page-locked memory from torch's caching host allocator (pageable
    D2H copies run at ~12 GB/s here, pinned ones at the PCIe rate: the 94 MB panel of configs[3] takes 8 ms vs 2);
    the array keeps its block alive and the allocator reuses it once the frame built on it is dropped."""
    import torch
    if t.numel() * t.element_size() < (1 << 20) or not t.is_cuda:
        return t.cpu().numpy()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    if wait:
        torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def _region_key_column(rid, n_time, weights):
    """The region table's id column for a full region-major panel (every region `n_time` rows), or None when `_merge_regions`'
    lookup does not apply (no table given, an index that is not unique and ascending, region ids out of order or not in it)."""
    if weights is None:
        return None
    gr = weights.georegions
    idx = gr.shp.index
    if not (idx.is_unique and idx.is_monotonic_increasing) or (len(rid) > 1 and not bool((rid[1:] > rid[:-1]).all())):
        return None
    upos = idx.get_indexer(rid)
    if (upos < 0).any():
        return None
    return np.repeat(gr.shp[gr.regionid].to_numpy()[upos], n_time)


def _assemble_frame(res, names, region_ids, labels, weights, merge_with=None) -> pd.DataFrame:
    """Long frame + NaN-row policy (`spatial.py:136-154`).  res: [K, R, P], a numpy array or an HBM tensor.
    ``merge_with``: the weights whose region table the caller would merge the frame with next (`aggregate_dataset`): a full panel
    then carries the table's id column straight away (frame.attrs["_merged"]) instead of positions that `_merge_regions` scans
    again — a million-row daily panel spends more time in those passes than in the kernels.

    The rows to keep are found on the [R, P] panel and only those are materialised — on the GPU when the
    panel is still there (mask, compaction, then one download of the kept values).  The reference builds
    the full R*P frame and drops rows (`spatial.py:144-153`); at configs[3] (3,600 regions x 251 years x
    13 bins) that pandas work costs ~100x the GPU time of the whole job."""
    names = list(names)
    rid = np.asarray(region_ids)
    zero_mask = None
    if getattr(weights, "zero_weight", "area") == "nan":
        zero_regions = _zero_weight_regions(weights.weights)
        if zero_regions:
            zero_mask = np.isin(rid, np.fromiter(zero_regions, dtype=rid.dtype, count=len(zero_regions)))
    if isinstance(res, np.ndarray):
        n_regions, n_time = res.shape[1], res.shape[2]
        ok = ~np.isnan(res).any(axis=0) if names else np.ones((n_regions, n_time), dtype=bool)      # [R, P]
        if zero_mask is not None:
            ok = ok | zero_mask[:, None]
        ri, ti = np.nonzero(ok)                                           # row-major: region, then time, like repeat/tile
        vals = [res[k][ri, ti] for k in range(len(names))]
    else:
        import torch
        n_regions, n_time = res.shape[1], res.shape[2]
        ok = ~torch.isnan(res).any(dim=0) if names else torch.ones((n_regions, n_time), dtype=torch.bool, device=res.device)
        if zero_mask is not None:
            ok = ok | torch.as_tensor(zero_mask, device=res.device)[:, None]
        n_ok = ok.sum()                                 # (queued behind the kernels; read below)
        kept = _download(res.reshape(res.shape[0], -1), wait=False) if names else None      # ... and the panel's way home, should it be full
        # the key and time columns of a full panel do not depend on the results: built while the kernels still run
        key = _region_key_column(rid, n_time, merge_with)
        if key is None:
            cols = {"region_id": np.repeat(rid, n_time)}
        else:                                           # aggregate_dataset: the merge with the region table is a lookup per region
            cols = {merge_with.georegions.regionid: key}
        cols["time"] = np.tile(_label_values(labels), n_regions)
        if int(n_ok) == n_regions * n_time:             # (reading it waits for the stream: the panel's copy is over too)
            # nothing dropped (the usual case): repeat / tile, like the reference
            for k, nm in enumerate(names):
                cols[nm] = kept[k]
            df = pd.DataFrame(cols, copy=False)
            if key is not None:
                df.attrs["_merged"] = True
            return df
        flat = ok.reshape(-1).nonzero().squeeze(1)
        kept = _download(res.reshape(res.shape[0], -1).index_select(1, flat)) if names else None
        flat = _download(flat)
        vals = [kept[k] for k in range(len(names))]
        ri, ti = np.divmod(flat, n_time)
    cols = {"region_id": rid[ri], "time": _label_values(labels)[ti]}
    for nm, v in zip(names, vals):
        cols[nm] = v
    return pd.DataFrame(cols, copy=False)


class SpatialAggregator:
    """Weighted regional average of temporally-reduced data (`spatial.py:37-154`)."""

    def __init__(self, dataset: Union[list, Dataset], weights, names: Union[str, List[str]] = "climate"):
        self.dataset = dataset if isinstance(dataset, list) else [dataset]
        self.grid = weights.grid
        self.weights_obj = weights
        self.weights = weights.weights
        self.names = [names] if isinstance(names, str) else list(names)
        self.zero_weight = getattr(weights, "zero_weight", "area")

    def compute(self, npartitions: int = None) -> pd.DataFrame:
        x, labels = _aligned_cells(self.dataset)                      # [K, n_cells, P] float64 in HBM, P = union of the labels
        csr, region_ids = eng.get_csr(self.weights_obj, self.dataset[0], device=x.device)
        _, _, res = csr.wavg(x)
        return _assemble_frame(res, self.names, region_ids, labels, self.weights_obj)


def aggregate_space(dataset_dict: Dict[str, Dataset], weights, npartitions=None, **kwargs) -> pd.DataFrame:
    """`aggregate_space` (`aggregate.py:165-198`)."""
    return SpatialAggregator(list(dataset_dict.values()), weights, names=list(dataset_dict.keys())).compute()


# --------------------------------------------------------------------------------------
# the whole path
# --------------------------------------------------------------------------------------
def _merge_regions(df: pd.DataFrame, weights) -> pd.DataFrame:
    """`aggregate.py:276-280`: ``shp[[regionid]].merge(df, left_index=True, right_on="region_id")`` minus the
    key column.  When the region table's index is unique and ascending and the panel is region-major (it
    always is here), that merge is a lookup: same rows, order, index and columns without pandas' generic join
    (which takes 60 ms on the 900 k-row panel of configs[3], ten times the kernels)."""
    gr = weights.georegions
    shp = gr.shp
    idx = shp.index
    rid = df["region_id"].to_numpy()
    if (idx.is_unique and idx.is_monotonic_increasing and gr.regionid not in df.columns and isinstance(df.index, pd.RangeIndex)
            and (len(rid) < 2 or bool((rid[1:] >= rid[:-1]).all()))):
        # region-major panel: look each region up once, then repeat its id over its run of rows
        starts = np.concatenate(([0], np.flatnonzero(rid[1:] != rid[:-1]) + 1)) if len(rid) else np.zeros(0, dtype=np.int64)
        runs = np.diff(np.concatenate((starts, [len(rid)])))
        upos = idx.get_indexer(rid[starts])
        cols = {gr.regionid: np.repeat(shp[gr.regionid].to_numpy()[upos], runs)}
        if (upos >= 0).all():
            for c in df.columns:
                if c != "region_id":
                    cols[c] = df[c].to_numpy()
            return pd.DataFrame(cols, index=df.index, copy=False)
    return shp[[gr.regionid]].merge(df, left_index=True, right_on="region_id").drop(columns="region_id")


def panel_arrays(weights, dataset: Dataset, aggregator_dict, engine: str = "auto"):
    """The numeric core of aggregate_dataset: -> (res [K,R,P] HBM tensor, names, region_ids,
    labels).  One fused pass when every column shares its group frequencies; otherwise the
    temporal outputs are gathered and reduced together so the validity mask stays shared
    across ALL names (`spatial.py:114-119`)."""
    import torch
    resolve_engine(engine)
    tindex = _labels_of(dataset)
    order, fused_cols, staged, names = _lower_all(aggregator_dict)
    _guard_week(fused_cols, tindex)
    csr, region_ids = eng.get_csr(weights, dataset, device=eng.dataset_device(dataset))
    if not staged:
        groups = eng.plan_groups(tindex, fused_cols)
        if len(groups) == 1 and [c.key for c in groups[0][0]] == names:
            cols, ib, ob, labels = groups[0]
            try:
                pr = eng.run_fused_pass(_time_major(dataset), cols, ib, ob, csr=csr, want_cells=False)
            except hip.HipUnsupported:
                pr = []
            if len(pr) == 1 and pr[0].panel is not None:
                return pr[0].panel["res"], names, region_ids, labels
    tdict = aggregate_time(dataset, weights, aggregator_dict, engine=engine)
    x, labels = _aligned_cells([tdict[nm] for nm in names])
    _, _, res = csr.wavg(x)
    return res, names, region_ids, labels


def aggregate_dataset(weights, dataset: Dataset = None, aggregator_dict=None, dataset_dict=None,
                      engine: str = "auto", **kwargs) -> pd.DataFrame:
    """`aggregate_dataset` (`aggregate.py:210-282`): [regionid, time, <output columns>]."""
    if dataset is None:
        raise ValueError("No dataset provided.")
    stale = {k: kwargs.pop(k) for k in _DEPRECATED_CLUSTER_KWARGS if k in kwargs}
    if stale:
        warnings.warn(
            f"aggregate_dataset no longer builds a Dask cluster; {sorted(stale)} is/are ignored. "
            "This engine runs on the GPU; no dask client is involved.", DeprecationWarning, stacklevel=2)
    if aggregator_dict is None and kwargs:
        aggregator_dict = kwargs
    if aggregator_dict is not None:
        res, names, region_ids, labels = panel_arrays(weights, dataset, aggregator_dict, engine)
        merge_ok = weights.georegions.regionid not in list(names) + ["time", "region_id"]
        df = _assemble_frame(res, names, region_ids, labels, weights, merge_with=weights if merge_ok else None)
        if df.attrs.pop("_merged", False):
            return df
    else:
        if dataset_dict is None:
            dataset_dict = {"variable": dataset}
        df = aggregate_space(dataset_dict, weights)
    return _merge_regions(df, weights)
