"""Host scheduler of the HIP engine: spec lowering, plan cache, device residency, CSR cache.

Replaces, for ``engine="hip"``, the lazy xarray/dask graph the reference builds in
`aggregate_time` (`aggfly/aggregate/aggregate.py:101-162`) and runs at
`aggfly/aggregate/spatial.py:125`:

* ``lower_spec`` walks one output name's step list exactly like the reference's
  interpreter loop (same fan-out, same key names, same errors) but symbolically, and emits
  *column programs* ``inner reducer -> transform -> outer reducer``;
* columns of all names that share their two group frequencies are packed into one fused
  pass (``hip.FusedPlan``) that reads the raw cube ONCE — the reference re-reads the raw
  data for every output name (`aggregate.py:133`);
* a step list that does not fit the two-level shape (three 'aggregate' levels, a transform
  on raw data or after the outer level) runs *staged*: the same GPU kernels, one step at a time,
  with the intermediates kept in HBM (the element-wise transforms through `hip.transform`).

Nothing here computes on the CPU; without the HIP library or a GPU every entry point raises.
"""
from __future__ import annotations

import hashlib
import os
import threading
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import pandas as pd

from . import hip
from .dataarray import _is_torch
from .dataset import Dataset
from .timegroups import resample_groups, translate_groupby

STAT_CALCS = ("mean", "sum", "min", "max", "nanmean")
THR_CALCS = ("dd", "bins", "sine_dd")
HIP_CALCS = frozenset(STAT_CALCS + THR_CALCS)           # NUMBA_CALCS, nb_kernels.py:34
FUSED_OUTER = ("sum", "mean", "min", "max", "dd", "bins")
MAX_COLS_PER_PASS = 16


class config:
    """Engine switches (also read from the environment at import)."""
    #: never split an outer period across workgroups: per-cell sums then run in exactly the
    #: reference's k-ascending order (bit-exact sums) at the cost of parallelism when there
    #: are few outer periods.
    exact_order = os.environ.get("AGGFLY_HIP_EXACT_ORDER", "0") == "1"
    #: float32 cubes: reproduce the reference's float32 intermediates (every step's output stored
    #: in its input dtype, `aggfly/aggregate/nb_kernels.py:257-262`; `np.power` with an int64
    #: exponent promotes to float64 under NumPy 2) instead of keeping float64 throughout
    match_reference_f32 = os.environ.get("AGGFLY_HIP_MATCH_F32", "0") == "1"
    #: kernel tuning arm (see include/aggfly_hip.h, afhip_plan_desc.tuning)
    tuning = int(os.environ.get("AGGFLY_HIP_TUNING", "0"))


# --------------------------------------------------------------------------------------
# spec lowering
# --------------------------------------------------------------------------------------
@dataclass
class AggStep:
    calc: str
    freq: str
    ddargs: Optional[tuple] = None      # one (t0, t1, flag) row, or None


@dataclass
class ColumnProg:
    key: str
    inner: Optional[AggStep] = None
    tf: Optional[tuple] = None          # ("pow", e) | ("hinge", knot)
    outer: Optional[AggStep] = None

    def levels(self):
        return (self.inner is not None) + (self.outer is not None)


def _agg_params(params):
    """Accept a params dict or a TemporalAggregator-like object (`aggregate.py:137-138`)."""
    if isinstance(params, dict):
        unknown = set(params) - {"calc", "groupby", "ddargs", "pre_compute"}
        if unknown:
            raise TypeError(f"TemporalAggregator got unexpected arguments {sorted(unknown)}")
        calc, groupby, ddargs = params["calc"], params["groupby"], params.get("ddargs")
        freq = translate_groupby(groupby)
    else:
        calc, freq, ddargs = params.calc, params.groupby, params.ddargs
    if calc not in HIP_CALCS:
        raise ValueError(f"unknown calc {calc!r}; supported: {sorted(HIP_CALCS)}")
    multi = ddargs is not None and np.array(ddargs).ndim > 1              # temporal.py:156-161
    if calc in THR_CALCS and ddargs is None:
        raise ValueError(f"calc {calc!r} needs ddargs")
    return calc, freq, ddargs, multi


def lower_spec(key: str, steps):
    """One output name -> (list[ColumnProg], fusable).

    Mirrors the loop at `aggregate.py:131-158`: 'aggregate' maps over the current column
    list; a multi-row ``ddargs`` fans out to ``{key}_{lo}_{hi}`` from the BASE key
    (`aggregate.py:148,299`) and refuses more than one input (`:144-147`); 'transform' fans
    out per column (`:150-157`).
    """
    cols = [ColumnProg(key)]
    fusable = True
    for kind, params in steps:
        if kind == "aggregate":
            calc, freq, ddargs, multi = _agg_params(params)
            rows = [tuple(float(v) for v in r) for r in np.atleast_2d(np.asarray(ddargs, dtype=float))] if ddargs is not None else [None]
            if multi and len(cols) > 1:
                raise ValueError("Cannot aggregate multiple datasets with multiple ddargs, "
                                 "e.g., multiple polynomials for multiple bins")
            new = []
            for c in cols:
                for r in rows:
                    step = AggStep(calc, freq, r)
                    n = ColumnProg(c.key, c.inner, c.tf, c.outer)
                    if n.inner is None and n.tf is None:
                        n.inner = step
                    elif n.outer is None and n.inner is not None:
                        n.outer = step
                        if calc not in FUSED_OUTER:
                            fusable = False
                    else:
                        fusable = False
                    new.append(n)
            if multi:
                raw = np.atleast_2d(np.asarray(ddargs, dtype=object))
                for n, x in zip(new, raw):
                    n.key = f"{key}_{x[0]}_{x[1]}"                         # aggregate.py:299
            cols = new
        elif kind == "transform":
            new = []
            for c in cols:
                if "exp" in params:
                    exp = params["exp"]
                    if not isinstance(exp, list):
                        exp = [exp]
                    items = [(f"{c.key}_{e}", ("pow", float(e))) for e in exp[0]]   # aggregate.py:54-63
                elif "inter" in params:
                    items = [(c.key, ("inter", params["inter"]))]
                elif "spline" in params.get("transform", ""):
                    items = [(f"{c.key}_spline1", None), (f"{c.key}_spline2", ("hinge", 20.0))]
                else:
                    raise ValueError("No valid transform argument provided.")
                for k2, tf in items:
                    n = ColumnProg(k2, c.inner, c.tf, c.outer)
                    if tf is not None:
                        if n.inner is None or n.outer is not None or n.tf is not None:
                            fusable = False
                        else:
                            n.tf = tf
                    new.append(n)
            cols = new
        else:
            raise ValueError(f"unknown step type {kind!r} (expected 'aggregate' or 'transform')")
    for c in cols:
        if c.inner is None:
            fusable = False
    return cols, fusable


# --------------------------------------------------------------------------------------
# device residency and caches
# --------------------------------------------------------------------------------------
def device_cube(dataset: Dataset):
    """The dataset's (time, lat, lon) cube as an HBM tensor (uploaded if still on the host)."""
    import torch
    hip.require_gpu()
    d = dataset.cube()
    if not _is_torch(d):
        if d.dtype not in (np.float32, np.float64):
            d = d.astype(np.float64)
        d = torch.from_numpy(d)
    if not d.is_cuda:
        d = d.cuda(non_blocking=True)
    if d.dtype not in (torch.float32, torch.float64):
        d = d.to(torch.float64)
    return d.contiguous()


def dataset_device(dataset) -> int:
    """Index of the GPU that holds (or, for a host array, will receive: the calling thread's current device) the dataset's
    cube: the device its plans and weight tables must be created on."""
    d = getattr(getattr(dataset, "da", None), "data", None)
    if d is not None and _is_torch(d) and d.is_cuda:
        return hip._device_index(d)
    return hip._device_index(None)


def _hash(*arrs) -> str:
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


_PLAN_CACHE: dict = {}
_PLAN_CACHE_MAX = 32
_PLAN_CACHE_BYTES = 16 << 30        # plans own their scratch in HBM: bound what the cache pins


def get_plan(T, n_cells, dtype_code, ib, ob, columns, exact_order=None, tuning=None, device=None) -> hip.FusedPlan:
    """``device``: the GPU of the cube the plan will run on (its tables and scratch live there: a plan is bound to the device
    it was created on, include/aggfly_hip.h "Devices"); part of the cache key, so that rank r / a worker thread on card r
    never picks up card 0's plan."""
    exact = config.exact_order if exact_order is None else exact_order
    tune = config.tuning if tuning is None else tuning
    dev = hip._device_index(device)
    # the second cube of an 'inter' column is bound per run (hip.FusedPlan.bind_inter), so it is not part of the key
    ckey = (dev, T, n_cells, dtype_code, _hash(ib, ob), repr([{k: v for k, v in c.items() if k != "inter"} for c in columns]), exact, tune)
    with _CACHE_LOCK:
        p = _PLAN_CACHE.get(ckey)
    if p is not None:
        return p
    # built OUTSIDE the cache lock (table uploads, occupancy queries, the chunk search): the first calls of the threads / devices of
    # a dask-style pool do not queue up behind one another
    p = hip.FusedPlan(T, n_cells, dtype_code, ib, ob, columns, exact_order=exact, tuning=tune, device=dev)
    with _CACHE_LOCK:
        hit = _PLAN_CACHE.get(ckey)
        if hit is not None:                             # another thread built the same plan meanwhile: keep one
            return hit
        held = sum(q.scratch_bytes() for q in _PLAN_CACHE.values())
        while _PLAN_CACHE and (len(_PLAN_CACHE) >= _PLAN_CACHE_MAX or held + p.scratch_bytes() > _PLAN_CACHE_BYTES):
            old = _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
            held -= old.scratch_bytes()
        _PLAN_CACHE[ckey] = p
    return p


def weight_triplets(wdf: pd.DataFrame, cell_ids: np.ndarray):
    """`_weight_triplets` (`aggfly/aggregate/spatial.py:157-178`), vectorised: rows = rank of
    ``index_right`` among the sorted unique ids, cols = position of ``cell_id`` in the grid's
    cell order, entries whose cell is absent from the grid dropped."""
    region_ids = np.sort(wdf["index_right"].unique())
    rows = np.searchsorted(region_ids, wdf["index_right"].to_numpy())
    cell_ids = np.asarray(cell_ids)
    cid = wdf["cell_id"].to_numpy()
    if np.array_equal(cell_ids, np.arange(len(cell_ids))):
        keep = (cid >= 0) & (cid < len(cell_ids))
        cols = cid
    else:
        order = np.argsort(cell_ids, kind="stable")
        pos = np.searchsorted(cell_ids[order], cid)
        pos = np.clip(pos, 0, len(cell_ids) - 1)
        keep = cell_ids[order][pos] == cid
        cols = order[pos]
    w = wdf["weight"].to_numpy(dtype=float)
    return rows[keep].astype(np.int64), np.asarray(cols)[keep].astype(np.int64), w[keep], region_ids


_CSR_CACHE: dict = {}


def get_csr(weights, dataset: Dataset, device=None):
    """CSR for ``weights`` laid over ``dataset``'s grid, cached per (device, table, grid); ``device``: the GPU that holds the
    cube (default: the calling thread's current device).

    The weights' ``cell_id`` lives on the +-180-sorted grid.  A 0-360 dataset is NOT
    re-sorted in memory (the reference does, `spatial.py:60`): the longitude permutation is
    folded into the CSR column indices instead, so the cube is streamed as stored.
    """
    wdf = weights.weights
    if wdf is None:
        raise ValueError("weights have no table; call calculate_weights() / pass table=")
    ny, nx = len(dataset.latitude), len(dataset.longitude)
    order, _ = dataset.lon_order_to_180()
    # keyed on the table object's identity; a finalizer evicts the entry when the table dies,
    # so a recycled id() can never serve a stale CSR
    ck = getattr(weights.grid, "_cell_key", None)
    if ck is not None and ck[0] == id(weights.grid.cell_id):
        cell_key = ck[1:]
    else:
        # (a copied grid, or a cell_id set by hand: hash it once and remember the result for as long as it is this very array —
        # 1 ms per call on a 310 k-cell grid otherwise)
        cell_key = (_hash(np.asarray(weights.grid.cell_id)),)
        try:
            weights.grid._cell_key = (id(weights.grid.cell_id),) + cell_key
        except AttributeError:
            pass
    dev = hip._device_index(device)
    ckey = (dev, id(wdf), len(wdf), ny, nx, _hash(order), cell_key)
    with _CACHE_LOCK:
        hit = _CSR_CACHE.get(ckey)
    if hit is not None:
        return hit
    try:
        weakref.finalize(wdf, _CSR_CACHE.pop, ckey, None)
    except TypeError:
        ckey = None      # not weak-referenceable: do not cache
    cell_ids = np.asarray(weights.grid.cell_id)
    if len(cell_ids) != ny * nx:
        raise ValueError(f"weights grid has {len(cell_ids)} cells but the dataset grid has {ny}x{nx}; "
                         "clip the dataset to the same regions as the weights")
    rows, cols, w, region_ids = weight_triplets(wdf, cell_ids)
    iy, ixs = np.divmod(cols, nx)
    cols_mem = iy * nx + order[ixs]            # sorted-grid position -> position in the stored cube
    csr = hip.CSR(rows, cols_mem, w, len(region_ids), ny * nx, device=dev)
    if ckey is not None:
        with _CACHE_LOCK:
            if ckey in _CSR_CACHE:                      # another thread built the same table meanwhile: keep one
                return _CSR_CACHE[ckey]
            if len(_CSR_CACHE) >= 8:
                _CSR_CACHE.pop(next(iter(_CSR_CACHE)))
            _CSR_CACHE[ckey] = (csr, region_ids)
    return csr, region_ids


# --------------------------------------------------------------------------------------
# fused execution
# --------------------------------------------------------------------------------------
def _is_f32(obj) -> bool:
    obj = getattr(obj, "da", obj)
    return str(getattr(obj, "dtype", "")).endswith("float32")


def inter_time_major(other, G1: int, ny: int, nx: int, device):
    """The second array of an 'inter' transform (`Dataset.interact`, `aggfly/dataset/dataset.py:483-518`) as an HBM tensor
    ``[G1, ny, nx]``.  Like the reference it must have the shape of the data it multiplies, else the same
    ``AssertionError``: a Dataset / labelled array is transposed by dimension name first (any order works); a bare array
    must be (time, latitude, longitude) of the inner level's output — the layout the reference's compiled engine leaves
    its step outputs in (`nb_kernels.py:293`), and the one the staged path of this engine has at the same point.  Only
    layout changes here: the product itself runs in the kernel."""
    import torch
    if isinstance(other, Dataset):
        other = other.da
    if hasattr(other, "dims") and hasattr(other, "data"):
        if set(other.dims) == {"latitude", "longitude", "time"}:
            other = other.transpose("time", "latitude", "longitude")
        other = other.data
    assert tuple(other.shape) == (G1, ny, nx), f"inter array has shape {tuple(other.shape)}, the data it multiplies {(G1, ny, nx)}"
    if not _is_torch(other):
        other = np.asarray(other)
        if other.dtype not in (np.float32, np.float64):
            other = other.astype(np.float64)
        other = torch.from_numpy(np.ascontiguousarray(other))
    other = other.to(device, non_blocking=True)
    if other.dtype not in (torch.float32, torch.float64):
        other = other.to(torch.float64)
    return other.contiguous()


def _column_dict(c: ColumnProg, f32_rules: bool = False) -> dict:
    d = {"inner": c.inner.calc}
    if f32_rules:
        # dtype walk of the reference's numba path on a float32 cube: an aggregate step stores its
        # output in its input's dtype; np.power with an int64 exponent promotes to float64
        # (NumPy 2 promotion); the hinge stays float32; float32 times a float32 array stays float32
        r = hip.ROUND_INNER
        is64 = c.tf is not None and (c.tf[0] == "pow" or (c.tf[0] == "inter" and not _is_f32(c.tf[1])))
        if c.tf is not None and (c.tf[0] == "hinge" or (c.tf[0] == "inter" and not is64)):
            r |= hip.ROUND_HINGE
        if c.outer is not None and not is64:
            r |= hip.ROUND_FINAL
        d["rounding"] = r
    if c.inner.ddargs is not None:
        d["inner_args"] = c.inner.ddargs
        if c.inner.calc == "sine_dd" and c.inner.ddargs[2] not in (0.0, 1.0):
            raise ValueError("Invalid ddargs[2] value")                    # temporal.py:324
    if c.tf is not None and c.tf[0] == "inter":
        d["transform"], d["inter"] = c.tf
    elif c.tf is not None:
        d["transform"], d["transform_arg"] = c.tf
    if c.outer is not None:
        d["outer"] = c.outer.calc
        if c.outer.ddargs is not None:
            d["outer_args"] = c.outer.ddargs
    else:
        d["outer"] = "identity"
    return d


@dataclass
class PassResult:
    keys: list
    labels: object
    plan: hip.FusedPlan
    cells: object = None        # [K, P, C] device tensor when materialised
    panel: dict = None          # num/den/res when the whole path ran fused


def plan_groups(time_index, cols):
    """Group fusable columns by their (inner freq, outer freq); -> list of
    (cols, ib, ob, labels).

    When every inner group holds exactly one step (daily data grouped by 'date') and the
    inner reducer is a plain statistic with no transform, the inner value IS the raw value
    (v/1 = v, NaN stays NaN), so the column is rewritten to a single level with the outer
    reducer applied straight to the raw steps: its thresholds then run in the streaming loop
    instead of the per-group epilogue.  Results are identical by construction."""
    level1 = {}
    for c in cols:
        if c.inner.freq not in level1:
            level1[c.inner.freq] = resample_groups(time_index, c.inner.freq)
    groups = {}
    for c in cols:
        ib, lab1 = level1[c.inner.freq]
        f2 = c.outer.freq if c.outer else None
        unit = f2 is not None and c.tf is None and c.inner.calc in STAT_CALCS and len(ib) > 1 and bool(np.all(np.diff(ib) == 1))
        groups.setdefault((c.inner.freq, f2, unit), []).append(c)
    out = []
    for (f1, f2, unit), cs in groups.items():
        ib, lab1 = level1[f1]
        if f2 is None:
            ob, labels = np.arange(len(ib), dtype=np.int64), lab1
        else:
            ob, labels = resample_groups(lab1, f2)
        if unit:
            cs = [ColumnProg(c.key, AggStep(c.outer.calc, c.outer.freq, c.outer.ddargs), None, None) for c in cs]
            ib, ob = ob, np.arange(len(ob), dtype=np.int64)
        for i in range(0, len(cs), MAX_COLS_PER_PASS):
            out.append((cs[i:i + MAX_COLS_PER_PASS], ib, ob, labels))
    return out


# Re-entrancy (the reference's kernels are called concurrently from dask's thread pool, `nb_kernels.py:271-305`).
# Two locks, neither global over the runs:
#   * ``_CACHE_LOCK`` guards the two caches (lookup / insert / evict) — held for microseconds: plans and tables are built outside it;
#   * every cached plan has its own lock (``hip.FusedPlan.lock``), held while ONE call binds its second cubes and enqueues its
#     kernel sequence: a plan handle owns scratch in HBM and must not be entered twice at once (include/aggfly_hip.h).
# Calls on DIFFERENT plans (other shapes, other specs, other devices) enqueue concurrently; calls on the same plan enqueue one
# after the other and stay in order on the stream they share, so no device synchronisation is needed.  Threads that put the
# SAME plan on different streams would race on its scratch: give them `exact_order` / `tuning` variants or separate processes.
_CACHE_LOCK = threading.RLock()


def run_fused_pass(cube, cols, ib, ob, csr=None, want_cells=True, exact_order=None):
    """Run one fused pass; splits it if the library says the pass is too wide."""
    return _run_fused_pass(cube, cols, ib, ob, csr, want_cells, exact_order)


def _run_fused_pass(cube, cols, ib, ob, csr=None, want_cells=True, exact_order=None):
    T = int(cube.shape[0])
    n_cells = int(cube[0].numel()) if T else int(np.prod(cube.shape[1:]))
    code = hip._dtype_code(cube)
    f32_rules = config.match_reference_f32 and code == hip.F32
    cdicts = [_column_dict(c, f32_rules) for c in cols]
    try:
        plan = get_plan(T, n_cells, code, ib, ob, cdicts, exact_order, device=cube.device)
    except hip.HipUnsupported:
        if len(cols) == 1:
            raise
        h = len(cols) // 2
        return _run_fused_pass(cube, cols[:h], ib, ob, None, True, exact_order) + \
            _run_fused_pass(cube, cols[h:], ib, ob, None, True, exact_order)
    inters = {j: inter_time_major(cd["inter"], len(ib) - 1, int(cube.shape[1]), int(cube.shape[2]), cube.device)
              for j, cd in enumerate(cdicts) if cd.get("transform") == "inter"}
    with plan.lock:                     # bind + enqueue as one step: another thread's call on this plan binds its own second cubes
        for j, other in inters.items():
            plan.bind_inter(j, other)
        if csr is not None:
            out = plan.run(cube, csr, want_cells=want_cells)
            return [PassResult([c.key for c in cols], None, plan, out.get("cells"), out)]
        return [PassResult([c.key for c in cols], None, plan, plan.run_temporal(cube), None)]
