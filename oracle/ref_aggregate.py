"""Oracle: spec-DSL interpreter and the end-to-end panel.  TEST INFRASTRUCTURE.

Restates `aggfly/aggregate/aggregate.py:36-303` on a minimal in-memory dataset:

* `aggregate_time` (`:101-162`): per output name, a list ``data`` and parallel ``keys``;
  every name restarts from the raw data (`:133`); 'aggregate' maps over ``data``
  (`:139-140`); multi-dd fans out to ``{key}_{lo}_{hi}`` (`:143-148`, `:299`) and refuses
  more than one input (`:144-147`); 'transform' fans out (`:150-157`);
* `transform_dataset` (`:36-78`): ``exp`` -> one array per exponent of ``exp[0]`` named
  ``{key}_{e}`` (`:54-63`, `np.power` per `aggfly/dataset/dataset.py:527-543`); ``inter``
  -> element-wise product (`dataset.py:547-563`); ``spline`` -> (x, (x>20)*(x-20)) named
  ``{key}_spline1/2`` (`aggregate.py:70-73`, `dataset.py:475-481`);
* `aggregate_dataset` (`:210-282`): temporal, then spatial on the +-180-sorted grid
  (`aggfly/aggregate/spatial.py:60`, `aggfly/dataset/grid_utils.py:16-73`), then the merge
  onto the shapefile's region ids (`aggregate.py:276-280`).
"""
from __future__ import annotations

import copy
import warnings
from dataclasses import dataclass, field

import numpy as np
import pandas as pd

from . import ref_temporal as rt
from .ref_calendar import OracleCFIndex
from .ref_spatial import spatial_compute

_DEPRECATED_CLUSTER_KWARGS = ("n_workers", "threads_per_worker", "processes", "memory_limit", "cluster_args")


@dataclass
class ODataset:
    """values[T, NY, NX] (time-major), time axis, coordinates, longitude convention."""
    values: np.ndarray
    time: object
    latitude: np.ndarray
    longitude: np.ndarray
    lon_is_360: bool = True
    history: list = field(default_factory=list)

    def deepcopy(self):
        return copy.deepcopy(self)

    def rescaled_to_180(self) -> "ODataset":
        """`Dataset.rescale_longitude` `dataset.py:419-440` + `array_lon_to_180`
        `grid_utils.py:52-73`: lon -> (lon+180)%360-180, then a stable sort by longitude."""
        if not self.lon_is_360:
            return self
        lon = (np.asarray(self.longitude, dtype=float) + 180) % 360 - 180
        order = np.argsort(lon, kind="stable")
        return ODataset(self.values[:, :, order], self.time, self.latitude, lon[order], False, list(self.history))


@dataclass
class OWeights:
    """What the hot path consumes of a GridWeights: the weights table
    (`aggfly/weights/grid_weights.py:194-196`), the grid's positional cell ids
    (`aggfly/dataset/grid.py:214-217`), the shapefile's region-id column
    (`aggregate.py:276-280`) and the zero-weight policy (`spatial.py:69`)."""
    table: pd.DataFrame
    cell_id: np.ndarray
    region_names: pd.Series           # index = shapefile row index, values = region id
    regionid: str = "geoid"
    zero_weight: str = "area"


class OTemporalAggregator:
    """`TemporalAggregator` `aggfly/aggregate/temporal.py:19-263`, arithmetic only."""

    def __init__(self, calc, groupby, ddargs=None, engine="numba"):
        self.calc = calc
        self.groupby = rt.translate_groupby(groupby)
        self.ddargs = ddargs
        self.multi_dd = ddargs is not None and np.array(ddargs).ndim > 1   # temporal.py:156-161
        self.engine = engine

    def execute(self, ds: ODataset):
        if self.groupby == "W" and isinstance(ds.time, OracleCFIndex):     # temporal.py:221-227
            raise NotImplementedError("groupby='week' is not supported on non-standard CF calendars")
        bounds, labels = rt.resample_groups(ds.time, self.groupby)
        fn = rt.numba_resample if self.engine == "numba" else rt.dask_resample
        out = fn(ds.values, bounds, self.calc, self.ddargs, self.multi_dd)
        hist = list(ds.history) + [self.groupby]
        if self.multi_dd:
            return [ODataset(np.ascontiguousarray(out[..., d]), labels, ds.latitude, ds.longitude,
                             ds.lon_is_360, list(hist)) for d in range(out.shape[-1])]
        return ODataset(out, labels, ds.latitude, ds.longitude, ds.lon_is_360, hist)


def transform_dataset(ds: ODataset, key: str, **kwargs):
    """`transform_dataset` `aggregate.py:36-78`."""
    def with_values(v):
        return ODataset(v, ds.time, ds.latitude, ds.longitude, ds.lon_is_360, list(ds.history))
    if "exp" in kwargs:
        exp = kwargs["exp"]
        if not isinstance(exp, list):
            exp = [exp]
        return ([with_values(np.power(ds.values, e)) for e in exp[0]],
                [f"{key}_{e}" for e in exp[0]])
    if "inter" in kwargs:
        other = kwargs["inter"]
        other = other.values if isinstance(other, ODataset) else np.asarray(other)
        assert ds.values.shape == other.shape
        return [with_values(np.multiply(ds.values, other))], [key]
    if "spline" in kwargs["transform"]:
        hinge = (ds.values > 20) * (ds.values - 20)
        return [ds, with_values(hinge)], [f"{key}_spline{x}" for x in (1, 2)]
    raise ValueError("No valid transform argument provided.")


def aggregate_time(dataset: ODataset, aggregator_dict=None, engine="numba", **kwargs):
    """`aggregate_time` `aggregate.py:101-162` -> {name: ODataset}."""
    if aggregator_dict is None:
        aggregator_dict = kwargs
    out = {}
    for key, steps in aggregator_dict.items():
        keys, data = [key], [dataset.deepcopy()]
        for kind, params in steps:
            if kind == "aggregate":
                agg = OTemporalAggregator(**params, engine=engine)
                data = [agg.execute(x) for x in data]
                if agg.multi_dd:
                    if len(data) > 1:
                        raise ValueError("Cannot aggregate multiple datasets with multiple ddargs")
                    data, keys = data[0], [f"{key}_{x[0]}_{x[1]}" for x in agg.ddargs]
            elif kind == "transform":
                nd, nk = [], []
                for d, k in zip(data, keys):
                    d2, k2 = transform_dataset(d, k, **params)
                    nd.extend(d2)
                    nk.extend(k2)
                data, keys = nd, nk
        out = out | dict(zip(keys, data))
    return out


def aggregate_space(dataset_dict: dict, weights: OWeights) -> pd.DataFrame:
    """`aggregate_space` `aggregate.py:165-198` + `SpatialAggregator.__init__/compute`."""
    names = list(dataset_dict)
    dsl = [dataset_dict[n].rescaled_to_180() for n in names]          # spatial.py:60
    # `xr.combine_by_coords` of one dataset per name (spatial.py:90-92) is an OUTER join on the time coordinate: the
    # output axis is the sorted union of the names' labels, and a name without a label there is NaN — which makes that
    # period invalid for every name (shared validity, spatial.py:114-119)
    time, pos = _union_time([d.time for d in dsl])
    arrs = {}
    for nm, d, at in zip(names, dsl, pos):
        T = d.values.shape[0]
        a = np.full((int(np.prod(d.values.shape[1:])), len(time)), np.nan)
        a[:, at] = np.asarray(d.values, dtype=np.float64).reshape(T, -1).T   # (cell, time), row-major lat x lon
        arrs[nm] = a
    tlabels = list(time) if not isinstance(time, pd.DatetimeIndex) else time.values
    return spatial_compute(arrs, tlabels, weights.table, weights.cell_id, weights.zero_weight)


def _union_time(axes):
    """Sorted union of time axes (all DatetimeIndex, or all on one CF calendar) and each axis' positions in it."""
    if all(isinstance(t, pd.DatetimeIndex) for t in axes):
        u = axes[0]
        for t in axes[1:]:
            u = u.union(t)
        return u, [u.get_indexer(t) for t in axes]
    first = axes[0]
    assert all(getattr(t, "calendar", None) == first.calendar for t in axes), "mixed calendars"
    key = lambda t: t.day_ordinal() * 24 + t.hour
    keys = [key(t) for t in axes]
    allk = np.unique(np.concatenate(keys))
    where = {}
    for t, k in zip(axes, keys):
        for i, kk in enumerate(k):
            where.setdefault(int(kk), (t, i))
    pick = [where[int(kk)] for kk in allk]
    OracleCFIndex = type(first)
    u = OracleCFIndex([t.year[i] for t, i in pick], [t.month[i] for t, i in pick], [t.day[i] for t, i in pick],
                      [t.hour[i] for t, i in pick], first.calendar)
    return u, [np.searchsorted(allk, k) for k in keys]


def aggregate_dataset(weights: OWeights, dataset: ODataset = None, aggregator_dict=None,
                      dataset_dict=None, engine="numba", **kwargs) -> pd.DataFrame:
    """`aggregate_dataset` `aggregate.py:210-282`."""
    if dataset is None:
        raise ValueError("No dataset provided.")
    stale = {k: kwargs.pop(k) for k in _DEPRECATED_CLUSTER_KWARGS if k in kwargs}
    if stale:
        warnings.warn(f"aggregate_dataset no longer builds a Dask cluster; {sorted(stale)} is/are ignored.",
                      DeprecationWarning, stacklevel=2)
    if aggregator_dict is None and kwargs:
        aggregator_dict = kwargs
    if aggregator_dict is not None:
        dataset_dict = aggregate_time(dataset, aggregator_dict, engine=engine)
    elif dataset_dict is None:
        dataset_dict = {"variable": dataset}
    df = aggregate_space(dataset_dict, weights)
    shp = weights.region_names.to_frame(weights.regionid)
    return shp.merge(df, left_index=True, right_on="region_id").drop(columns="region_id")
