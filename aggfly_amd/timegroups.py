"""Group bounds and output labels for the temporal stage (host side).

Mirrors `resample_groups` / `translate_groupby` of the reference
(`aggfly/aggregate/nb_kernels.py:80-115`, `aggfly/aggregate/temporal.py:441-456`): contiguous,
time-sorted bins that INCLUDE empty interior bins as zero-width ranges, so the kernels see
``bounds[g]..bounds[g+1]`` exactly as the numba kernels do.

A standard ``DatetimeIndex`` goes through the same pandas call the reference makes
(`nb_kernels.py:113-115`); CF calendars go through ``cfcalendar.resample_bins``.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from .cfcalendar import CFTimeIndex, resample_bins

_FREQ = {"date": "1D", "month": "ME", "year": "YE", "week": "W"}


def translate_groupby(groupby: str) -> str:
    """'date'|'month'|'year'|'week' -> pandas offset; unknown names raise KeyError
    (`temporal.py:441-456`)."""
    return _FREQ[groupby]


def as_time_index(time):
    """Normalise a time coordinate to DatetimeIndex or CFTimeIndex."""
    if isinstance(time, (pd.DatetimeIndex, CFTimeIndex)):
        return time
    arr = np.asarray(time)
    if arr.dtype.kind == "M":
        return pd.DatetimeIndex(arr)
    if arr.dtype == object and len(arr) and hasattr(arr[0], "calendar"):
        first = arr[0]
        return CFTimeIndex.from_fields([t.year for t in arr], [t.month for t in arr], [t.day for t in arr],
                                       [t.hour for t in arr], [getattr(t, "minute", 0) for t in arr],
                                       [getattr(t, "second", 0) for t in arr], calendar=first.calendar)
    return pd.DatetimeIndex(arr)


_GROUPS_CACHE: dict = {}


def resample_groups(tindex, freq: str):
    """-> (bounds int64[G+1], labels) matching ``da.resample(time=freq)`` bins.  The answer for a DatetimeIndex is
    cached on the index's values (a year of hourly stamps costs pandas ~0.5 ms per call, which sits in front of every
    kernel launch of a repeated job)."""
    if isinstance(tindex, pd.DatetimeIndex):
        key = (freq, len(tindex), hash(tindex.asi8.tobytes()), str(tindex.tz))
        hit = _GROUPS_CACHE.get(key)
        if hit is None:
            hit = _resample_groups(tindex, freq)
            if len(_GROUPS_CACHE) >= 64:
                _GROUPS_CACHE.pop(next(iter(_GROUPS_CACHE)))
            _GROUPS_CACHE[key] = hit
        return hit[0].copy(), hit[1]
    return _resample_groups(tindex, freq)


def _resample_groups(tindex, freq: str):
    if not tindex.is_monotonic_increasing:
        raise ValueError(
            "the temporal engine requires a monotonic-increasing time index "
            "(xarray's resample path enforces the same).")
    if isinstance(tindex, CFTimeIndex):
        counts, labels = resample_bins(tindex, freq)
        bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return bounds, labels
    counts = pd.Series(1, index=tindex).resample(freq).count()
    bounds = np.concatenate([[0], np.cumsum(counts.values)]).astype(np.int64)
    return bounds, pd.DatetimeIndex(counts.index)


def nest_bounds(inner_labels, freq_outer: str):
    """Outer bounds over inner groups: resample the inner level's labels at the outer
    frequency.  This is what the reference does when a second ('aggregate', ...) step
    runs on the first step's output (its time axis = the first step's labels)."""
    return resample_groups(inner_labels, freq_outer)
