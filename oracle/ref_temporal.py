"""Oracle: grouped temporal reducers.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates, on plain numpy arrays,

* the four numba kernels of the reference, `aggfly/aggregate/nb_kernels.py:121-251`
  (``numba_*`` functions below).  Each cell's arithmetic runs in exactly the reference's
  order (k ascending inside a group, float64 accumulators, result stored in the input
  dtype, `nb_kernels.py:257-268`); the loops over cells are vectorised, which does not
  change any per-cell result;
* their dask-path twins, `aggfly/aggregate/temporal.py:266-438` (``dask_*`` functions),
  i.e. the vectorised numpy reducers xarray calls once per resample group
  (`temporal.py:236-239`), with an empty resample bin reindexed to NaN;
* the group-bounds builder `resample_groups`, `nb_kernels.py:80-115`.

Arrays are time-major: ``cube[T, NY, NX]`` (`nb_kernels.py:280` transposes to this).
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from .ref_calendar import OracleCFIndex, cf_resample_groups

STAT_CODE = {"mean": 0, "sum": 1, "min": 2, "max": 3, "nanmean": 4}  # nb_kernels.py:33
FREQ = {"date": "1D", "month": "ME", "year": "YE", "week": "W"}       # temporal.py:456


def translate_groupby(groupby: str) -> str:
    """`temporal.py:441-456`: unknown names raise KeyError."""
    return FREQ[groupby]


def resample_groups(tindex, freq: str):
    """(bounds int64[G+1], labels) — `nb_kernels.py:80-115`.

    DatetimeIndex: the reference's own pandas call (`:113-115`).  CF calendars: the
    restated cftime grouping (`:100-110`) in ref_calendar.py.
    """
    if not tindex.is_monotonic_increasing:
        raise ValueError("numba engine requires a monotonic-increasing time index")
    if isinstance(tindex, OracleCFIndex):
        return cf_resample_groups(tindex, freq)
    counts = pd.Series(1, index=tindex).resample(freq).count()
    bounds = np.concatenate([[0], np.cumsum(counts.values)]).astype(np.int64)
    return bounds, pd.DatetimeIndex(counts.index)


# --------------------------------------------------------------------------- #
# numba-path kernels (nb_kernels.py:121-251), per-cell order preserved
# --------------------------------------------------------------------------- #
def numba_stat(cube: np.ndarray, bounds: np.ndarray, code: int) -> np.ndarray:
    """`_block_stat` `nb_kernels.py:121-155` -> out[G, NY, NX] in cube.dtype."""
    G = len(bounds) - 1
    shp = cube.shape[1:]
    out = np.empty((G,) + shp, dtype=cube.dtype)
    for g in range(G):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        n = np.zeros(shp, dtype=np.int64)
        s = np.zeros(shp, dtype=np.float64)
        mn = np.full(shp, np.inf)
        mx = np.full(shp, -np.inf)
        hasnan = np.zeros(shp, dtype=bool)
        for k in range(lo, hi):
            v = cube[k].astype(np.float64)
            isn = np.isnan(v)
            hasnan |= isn
            ok = ~isn
            s = np.where(ok, s + np.where(ok, v, 0.0), s)
            n += ok
            mn = np.where(ok & (v < mn), v, mn)
            mx = np.where(ok & (v > mx), v, mx)
        with np.errstate(invalid="ignore", divide="ignore"):
            if hi == lo:
                res = np.full(shp, np.nan)
            elif code == 4:
                res = np.where(n > 0, s / np.maximum(n, 1), np.nan)
            elif code == 0:
                res = np.where(hasnan, np.nan, s / np.maximum(n, 1))
            elif code == 1:
                res = np.where(hasnan, np.nan, s)
            elif code == 2:
                res = np.where(hasnan, np.nan, mn)
            else:
                res = np.where(hasnan, np.nan, mx)
        out[g] = res
    return out


def _ddargs2d(ddargs) -> np.ndarray:
    return np.atleast_2d(np.asarray(ddargs, dtype=np.float64))  # nb_kernels.py:294


def numba_dd(cube, bounds, ddargs) -> np.ndarray:
    """`_block_dd` `nb_kernels.py:158-179` -> out[G, NY, NX, D]."""
    dda = _ddargs2d(ddargs)
    G, D = len(bounds) - 1, dda.shape[0]
    shp = cube.shape[1:]
    out = np.empty((G,) + shp + (D,), dtype=cube.dtype)
    for g in range(G):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        for d in range(D):
            t0, t1 = dda[d, 0], dda[d, 1]
            base = t0 if dda[d, 2] == 0 else t1
            acc = np.zeros(shp)
            hasnan = np.zeros(shp, dtype=bool)
            for k in range(lo, hi):
                v = cube[k].astype(np.float64)
                isn = np.isnan(v)
                hasnan |= isn
                with np.errstate(invalid="ignore"):
                    m = (~isn) & (v > t0) & (v < t1)
                av = np.abs(np.where(m, v, base) - base)
                acc = np.where(m, acc + av, acc)
            out[g, ..., d] = np.where(hasnan | (hi == lo), np.nan, acc)
    return out


def numba_bins(cube, bounds, ddargs) -> np.ndarray:
    """`_block_bins` `nb_kernels.py:182-199`; a NaN value is out of range, never NaN."""
    dda = _ddargs2d(ddargs)
    G, D = len(bounds) - 1, dda.shape[0]
    shp = cube.shape[1:]
    out = np.empty((G,) + shp + (D,), dtype=cube.dtype)
    for g in range(G):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        for d in range(D):
            t0, t1 = dda[d, 0], dda[d, 1]
            c = np.zeros(shp)
            for k in range(lo, hi):
                v = cube[k].astype(np.float64)
                with np.errstate(invalid="ignore"):
                    c += ((v > t0) & (v < t1)).astype(np.float64)
            out[g, ..., d] = np.nan if hi == lo else c
    return out


def _sine_part_cooling(thr, tmin, tmax, tavg):
    """Cooling branch of `nb_kernels.py:227-236` for scalar threshold, array stats."""
    with np.errstate(invalid="ignore", divide="ignore"):
        rng = tmax - tmin
        a = np.arccos((2.0 * thr - tmax - tmin) / rng)
        mid = ((tavg - thr) * a + rng * np.sin(a) / 2.0) / np.pi
    part = np.zeros_like(tavg)
    c1 = thr <= tmin
    c2 = (~c1) & (thr < tmax) & (tmin < thr)
    part = np.where(c1, tavg - thr, part)
    part = np.where(c2, mid, part)
    return part


def _sine_part_heating(thr, tmin, tmax, tavg):
    """Heating branch of `nb_kernels.py:238-249`."""
    with np.errstate(invalid="ignore", divide="ignore"):
        alpha = (tmax - tmin) / 2.0
        r = (thr - tavg) / alpha
        at = np.arctan(r / np.sqrt(1.0 - r * r))
        mid = (1.0 / np.pi) * ((thr - tavg) * (at + np.pi / 2.0) + alpha * np.cos(at))
    part = np.zeros_like(tavg)
    c1 = thr >= tmax
    c2 = (~c1) & (thr < tmax) & (tmin < thr)
    part = np.where(c1, thr - tavg, part)
    part = np.where(c2, mid, part)
    return part


def numba_sine_dd(cube, bounds, ddargs) -> np.ndarray:
    """`_block_sine_dd` `nb_kernels.py:202-251` -> out[G, NY, NX, D]."""
    dda = _ddargs2d(ddargs)
    G, D = len(bounds) - 1, dda.shape[0]
    shp = cube.shape[1:]
    out = np.empty((G,) + shp + (D,), dtype=cube.dtype)
    for g in range(G):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        n = np.zeros(shp, dtype=np.int64)
        s = np.zeros(shp)
        tmax = np.full(shp, -np.inf)
        tmin = np.full(shp, np.inf)
        hasnan = np.zeros(shp, dtype=bool)
        for k in range(lo, hi):
            v = cube[k].astype(np.float64)
            isn = np.isnan(v)
            hasnan |= isn
            ok = ~isn
            s = np.where(ok, s + np.where(ok, v, 0.0), s)
            n += ok
            tmax = np.where(ok & (v > tmax), v, tmax)
            tmin = np.where(ok & (v < tmin), v, tmin)
        bad = hasnan | (n == 0)
        with np.errstate(invalid="ignore", divide="ignore"):
            tavg = s / np.maximum(n, 1)
        for d in range(D):
            kind = dda[d, 2]
            val = np.zeros(shp)
            for j in range(2):
                thr = dda[d, j]
                if kind == 0:
                    part = _sine_part_cooling(thr, tmin, tmax, tavg)
                    val = val + (part if j == 0 else -part)
                else:
                    part = _sine_part_heating(thr, tmin, tmax, tavg)
                    val = val + (-part if j == 0 else part)
            out[g, ..., d] = np.where(bad, np.nan, val)
    return out


NUMBA_FUNCS = {"dd": numba_dd, "bins": numba_bins, "sine_dd": numba_sine_dd}


def numba_resample(cube, bounds, calc, ddargs=None, multi_dd=False):
    """Driver `numba_resample` `nb_kernels.py:271-305` on a (T,NY,NX) array.

    Returns out[G,NY,NX] or, for multi-dd, out[G,NY,NX,D] (`:303-304` squeezes D
    otherwise).
    """
    if cube.ndim != 3:
        raise ValueError(f"numba engine expects 2 spatial dims, got {cube.ndim - 1}")
    if calc in STAT_CODE:
        return numba_stat(cube, bounds, STAT_CODE[calc])
    out = NUMBA_FUNCS[calc](cube, bounds, ddargs)
    return out if multi_dd else out[..., 0]


# --------------------------------------------------------------------------- #
# dask-path reducers (temporal.py:266-438), applied per resample group
# --------------------------------------------------------------------------- #
def _dask_dd_one(frame, dd):
    """`_dd` `temporal.py:266-289` on frame[t, NY, NX], reduced over axis 0."""
    with np.errstate(invalid="ignore"):
        return ((frame > dd[0]) * (frame < dd[1]) * np.absolute(frame - dd[int(dd[2])])).sum(axis=0)


def _dask_bins_one(frame, dd):
    """`_bins` `temporal.py:394-416` (integer sum of a bool mask)."""
    with np.errstate(invalid="ignore"):
        return ((frame > dd[0]) * (frame < dd[1])).sum(axis=0)


def _dask_sine_cdd(frame, thr):
    """`_sine_cdd` `temporal.py:328-350`."""
    nan_cells = np.where(np.isnan(frame).any(axis=0), np.nan, 1.0)
    tmax, tmin, tavg = frame.max(axis=0), frame.min(axis=0), frame.mean(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        case2 = np.where(thr <= tmin, tavg - thr, 0)
        ac = np.arccos((2 * thr - tmax - tmin) / (tmax - tmin))
        case3 = np.where((thr < tmax) & (tmin < thr),
                         ((tavg - thr) * ac + (tmax - tmin) * np.sin(ac) / 2) / np.pi, 0)
    return (case2 + case3) * nan_cells


def _dask_sine_hdd(frame, thr):
    """`_sine_hdd` `temporal.py:352-391`."""
    nan_cells = np.where(np.isnan(frame).any(axis=0), np.nan, 1.0)
    tmax, tmin, tavg = frame.max(axis=0), frame.min(axis=0), frame.mean(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        case2 = np.where(thr >= tmax, thr - tavg, 0)
        r = (thr - tavg) / ((tmax - tmin) / 2)
        at = np.arctan(r / np.sqrt(1 - r ** 2))
        case3 = np.where((thr < tmax) & (tmin < thr),
                         (1 / np.pi) * ((thr - tavg) * (at + np.pi / 2) + ((tmax - tmin) / 2) * np.cos(at)), 0)
    return (case2 + case3) * nan_cells


def _dask_sine_one(frame, dd):
    """`_sine_dd` `temporal.py:313-326`; bad flag raises ValueError."""
    if dd[2] == 0:
        return _dask_sine_cdd(frame, dd[0]) - _dask_sine_cdd(frame, dd[1])
    if dd[2] == 1:
        return _dask_sine_hdd(frame, dd[1]) - _dask_sine_hdd(frame, dd[0])
    raise ValueError("Invalid ddargs[2] value")


def dask_resample(cube, bounds, calc, ddargs=None, multi_dd=False):
    """`ds.resample(time=freq).reduce(func, **kw)` `temporal.py:236-239`.

    xarray calls the reducer once per non-empty group and reindexes empty bins to NaN.
    The result dtype follows numpy (bins -> integer counts, promoted to float only when
    an empty bin forces NaN).
    """
    G = len(bounds) - 1
    shp = cube.shape[1:]
    if calc in ("dd", "bins", "sine_dd"):
        dds = np.atleast_2d(np.asarray(ddargs, dtype=np.float64)) if multi_dd else [np.asarray(ddargs, dtype=np.float64)]
        one = {"dd": _dask_dd_one, "bins": _dask_bins_one, "sine_dd": _dask_sine_one}[calc]
        D = len(dds)
        out = np.full((G,) + shp + (D,), np.nan)
        for g in range(G):
            lo, hi = int(bounds[g]), int(bounds[g + 1])
            if hi > lo:
                for d in range(D):
                    out[g, ..., d] = one(cube[lo:hi].astype(np.float64), dds[d])
        return out if multi_dd else out[..., 0]
    fn = {"mean": np.mean, "sum": np.sum, "min": np.min, "max": np.max, "nanmean": np.nanmean}[calc]
    out = np.full((G,) + shp, np.nan, dtype=cube.dtype)
    for g in range(G):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        if hi > lo:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                out[g] = fn(cube[lo:hi], axis=0)
    return out
