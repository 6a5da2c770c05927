"""CF calendars without cftime: the time-axis half of the hot path's host logic.

The reference leans on xarray's CFTimeIndex for CMIP6-style calendars (``noleap``,
``360_day``, ``all_leap``; `aggfly/aggregate/nb_kernels.py:100-110`).  Neither cftime nor
xarray exists on the GPU box, so this module supplies the small part the path needs:

* ``CFDatetime`` / ``CFTimeIndex``: timestamps that know their calendar (the reference's
  tests check ``.calendar`` on the output panel's ``time`` values,
  `aggfly/tests/test_aggregate.py:533`);
* ``cf_range``: the equivalent of ``xr.date_range(..., calendar=..., use_cftime=True)``;
* ``decode_cf_time``: numeric "<units> since <epoch>" values -> CFTimeIndex (what xarray's
  decoder does when a store carries a non-standard calendar);
* ``resample_bins``: per-output-bin counts and labels for the freqs aggfly uses
  (``"1D"``, ``"ME"``, ``"YE"``; `aggfly/aggregate/temporal.py:456`), empty interior bins
  kept, labels at the bin start for days and at the period end for months / years.
"""
from __future__ import annotations

import re
from functools import total_ordering

import numpy as np

_MONTH_DAYS = {
    "noleap": (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31),
    "365_day": (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31),
    "all_leap": (31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31),
    "366_day": (31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31),
    "360_day": (30,) * 12,
}
STANDARD_CALENDARS = ("standard", "gregorian", "proleptic_gregorian")


def month_days(calendar: str):
    try:
        return _MONTH_DAYS[calendar]
    except KeyError:
        raise ValueError(f"unsupported CF calendar {calendar!r}; supported: {sorted(_MONTH_DAYS)}") from None


def year_days(calendar: str) -> int:
    return sum(month_days(calendar))


@total_ordering
class CFDatetime:
    """A timestamp on a fixed-length-year CF calendar."""

    __slots__ = ("year", "month", "day", "hour", "minute", "second", "calendar")

    def __init__(self, year, month, day, hour=0, minute=0, second=0, calendar="noleap"):
        md = month_days(calendar)
        if not (1 <= month <= 12 and 1 <= day <= md[month - 1]):
            raise ValueError(f"invalid date {year}-{month}-{day} on calendar {calendar}")
        self.year, self.month, self.day = int(year), int(month), int(day)
        self.hour, self.minute, self.second = int(hour), int(minute), int(second)
        self.calendar = calendar

    def _key(self):
        return (self.year, self.month, self.day, self.hour, self.minute, self.second)

    def __eq__(self, other):
        return isinstance(other, CFDatetime) and self.calendar == other.calendar and self._key() == other._key()

    def __lt__(self, other):
        if not isinstance(other, CFDatetime) or self.calendar != other.calendar:
            return NotImplemented
        return self._key() < other._key()

    def __hash__(self):
        return hash((self._key(), self.calendar))

    def isoformat(self):
        return f"{self.year:04d}-{self.month:02d}-{self.day:02d} {self.hour:02d}:{self.minute:02d}:{self.second:02d}"

    def __str__(self):
        return self.isoformat()

    def __repr__(self):
        return f"CFDatetime({self.isoformat()}, calendar={self.calendar!r})"


class CFTimeIndex:
    """Time axis on a CF calendar, stored as seconds since 0000-01-01 of that calendar."""

    def __init__(self, seconds, calendar: str):
        month_days(calendar)
        self.seconds = np.asarray(seconds, dtype=np.int64)
        self.calendar = calendar

    # ---- construction helpers ----
    @classmethod
    def from_fields(cls, year, month, day, hour=0, minute=0, second=0, calendar="noleap"):
        md = np.asarray(month_days(calendar))
        cum = np.concatenate([[0], np.cumsum(md)])
        year, month, day = (np.asarray(a, dtype=np.int64) for a in (year, month, day))
        days = year * cum[-1] + cum[month - 1] + (day - 1)
        secs = days * 86400 + np.asarray(hour, dtype=np.int64) * 3600 + np.asarray(minute, dtype=np.int64) * 60 \
            + np.asarray(second, dtype=np.int64)
        return cls(secs, calendar)

    # ---- field access ----
    def fields(self):
        md = np.asarray(month_days(self.calendar))
        cum = np.concatenate([[0], np.cumsum(md)])
        days, sod = np.divmod(self.seconds, 86400)
        year, doy = np.divmod(days, cum[-1])
        month = np.searchsorted(cum, doy, side="right")
        day = doy - cum[month - 1] + 1
        return year, month, day, sod

    def __len__(self):
        return len(self.seconds)

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            y, m, d, sod = CFTimeIndex(self.seconds[i:i + 1] if i >= 0 else self.seconds[[i]], self.calendar).fields()
            s = int(sod[0])
            return CFDatetime(int(y[0]), int(m[0]), int(d[0]), s // 3600, (s % 3600) // 60, s % 60, self.calendar)
        return CFTimeIndex(self.seconds[i], self.calendar)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __eq__(self, other):
        return isinstance(other, CFTimeIndex) and self.calendar == other.calendar and \
            np.array_equal(self.seconds, other.seconds)

    @property
    def is_monotonic_increasing(self) -> bool:
        return bool(np.all(np.diff(self.seconds) >= 0))

    def argsort(self):
        return np.argsort(self.seconds, kind="stable")

    # ---- label selection (what ``da.sel(time=...)`` does on an xarray CFTimeIndex) ----
    def _partial_span(self, label):
        """[lo, hi) in seconds for a partial date string ('2000', '2000-06', '2000-06-15', '2000-06-15T12', ...) — the
        whole year / month / day / hour it names — or the single instant of a `CFDatetime`."""
        if isinstance(label, CFDatetime):
            if label.calendar != self.calendar:
                raise ValueError(f"time_sel is on calendar {label.calendar!r}, the index on {self.calendar!r}")
            t = CFTimeIndex.from_fields([label.year], [label.month], [label.day], [label.hour], [label.minute], [label.second],
                                        calendar=self.calendar).seconds[0]
            return int(t), int(t) + 1
        m = re.fullmatch(r"\s*(-?\d{1,4})(?:-(\d{1,2})(?:-(\d{1,2})(?:[ T](\d{1,2})(?::(\d{1,2})(?::(\d{1,2}))?)?)?)?)?\s*", str(label))
        if not m:
            raise ValueError(f"cannot parse time selection {label!r} on a CF calendar")
        f = [int(g) if g is not None else None for g in m.groups()]
        md = month_days(self.calendar)
        y = f[0]
        mo, d, hh, mi, ss = (f[1] or 1), (f[2] or 1), (f[3] or 0), (f[4] or 0), (f[5] or 0)
        lo = int(CFTimeIndex.from_fields([y], [mo], [d], [hh], [mi], [ss], calendar=self.calendar).seconds[0])
        if f[1] is None:
            width = year_days(self.calendar) * 86400
        elif f[2] is None:
            width = md[mo - 1] * 86400
        elif f[3] is None:
            width = 86400
        elif f[4] is None:
            width = 3600
        elif f[5] is None:
            width = 60
        else:
            width = 1
        return lo, lo + width

    def sel_positions(self, time_sel):
        """Positions selected by ``time_sel``: a partial date string selects everything inside the year / month / day it
        names; a slice of two such labels is inclusive at both ends (xarray's partial-datetime-string indexing)."""
        if isinstance(time_sel, slice):
            lo = self._partial_span(time_sel.start)[0] if time_sel.start is not None else None
            hi = self._partial_span(time_sel.stop)[1] if time_sel.stop is not None else None
        else:
            lo, hi = self._partial_span(time_sel)
        keep = np.ones(len(self), dtype=bool)
        if lo is not None:
            keep &= self.seconds >= lo
        if hi is not None:
            keep &= self.seconds < hi
        return np.nonzero(keep)[0]

    def to_list(self):
        return list(self)

    def __repr__(self):
        n = len(self)
        ends = f"{self[0]} .. {self[-1]}" if n else ""
        return f"CFTimeIndex(n={n}, calendar={self.calendar!r}, {ends})"


def cf_range(start, periods: int, freq: str = "D", calendar: str = "noleap") -> CFTimeIndex:
    """``xr.date_range(start, periods=..., freq=..., calendar=..., use_cftime=True)`` for
    fixed steps: freq is ``"D"``, ``"h"``/``"H"`` or ``"<n>h"``/``"<n>D"``."""
    if isinstance(start, str):
        y, m, d = (int(x) for x in start.split("T")[0].split(" ")[0].split("-"))
    else:
        y, m, d = start
    m_ = re.fullmatch(r"(\d*)([DdHh])", freq)
    if not m_:
        raise ValueError(f"cf_range: unsupported freq {freq!r}")
    n = int(m_.group(1) or 1)
    step = n * (86400 if m_.group(2) in "Dd" else 3600)
    t0 = CFTimeIndex.from_fields([y], [m], [d], calendar=calendar).seconds[0]
    return CFTimeIndex(t0 + step * np.arange(periods, dtype=np.int64), calendar)


_UNIT_SECONDS = {"second": 1, "seconds": 1, "s": 1, "minute": 60, "minutes": 60, "hour": 3600, "hours": 3600,
                 "h": 3600, "day": 86400, "days": 86400, "d": 86400}


def decode_cf_time(values, units: str, calendar: str) -> CFTimeIndex:
    """CF "units since epoch" numbers -> CFTimeIndex (non-standard calendars only)."""
    m = re.fullmatch(r"\s*(\w+)\s+since\s+(\d{1,4})-(\d{1,2})-(\d{1,2})(?:[ T](\d{1,2}):(\d{1,2})(?::(\d{1,2})(?:\.\d*)?)?)?.*", units)
    if not m:
        raise ValueError(f"cannot parse CF time units {units!r}")
    unit = m.group(1).lower()
    if unit not in _UNIT_SECONDS:
        raise ValueError(f"unsupported CF time unit {unit!r}")
    y, mo, d = int(m.group(2)), int(m.group(3)), int(m.group(4))
    hh, mm, ss = (int(g) if g else 0 for g in m.group(5, 6, 7))
    epoch = CFTimeIndex.from_fields([y], [mo], [d], [hh], [mm], [ss], calendar=calendar).seconds[0]
    secs = np.round(np.asarray(values, dtype=np.float64) * _UNIT_SECONDS[unit]).astype(np.int64)
    return CFTimeIndex(epoch + secs, calendar)


def resample_bins(index: CFTimeIndex, freq: str):
    """(counts int64[G], labels CFTimeIndex[G]) for freq in {"1D", "ME", "YE"}.

    Bins run without gaps from the bin of the first timestamp to the bin of the last; an
    interior bin without data has count 0 (the reference zero-fills xarray's NaN count,
    `nb_kernels.py:104-109`).  ``"W"`` has no CF-calendar meaning and raises like the
    reference does (`aggfly/aggregate/temporal.py:221-227`).
    """
    if freq == "W":
        raise NotImplementedError(
            "groupby='week' is not supported on non-standard CF calendars (noleap/360_day/etc.): "
            "there is no calendar week. Use 'date', 'month', or 'year'.")
    if len(index) == 0:
        return np.zeros(0, dtype=np.int64), CFTimeIndex(np.zeros(0, dtype=np.int64), index.calendar)
    year, month, day, _ = index.fields()
    md = np.asarray(month_days(index.calendar))
    cum = np.concatenate([[0], np.cumsum(md)])
    if freq == "1D":
        ordinal = index.seconds // 86400
    elif freq == "ME":
        ordinal = year * 12 + (month - 1)
    elif freq == "YE":
        ordinal = year
    else:
        raise KeyError(freq)
    first = int(ordinal[0])
    nb = int(ordinal[-1]) - first + 1
    counts = np.bincount((ordinal - first).astype(np.int64), minlength=nb).astype(np.int64)
    o = first + np.arange(nb, dtype=np.int64)
    if freq == "1D":
        secs = o * 86400
    elif freq == "ME":
        y, m0 = np.divmod(o, 12)
        secs = (y * cum[-1] + cum[m0] + (md[m0] - 1)) * 86400          # last day of the month, 00:00
    else:
        secs = (o * cum[-1] + cum[11] + (md[11] - 1)) * 86400          # last day of the year, 00:00
    return counts, CFTimeIndex(secs, index.calendar)
