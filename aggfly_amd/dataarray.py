"""A small labelled array: the slice of xarray.DataArray the hot path's callers touch.

xarray is not installed here or on the GPU box.  The reference's API hands users
``Dataset.da`` (an ``xr.DataArray``) and its tests build inputs with
``xr.DataArray(data=..., dims=[...], coords={...})`` (`aggfly/tests/test_aggregate.py:44-52`).
``DataArray`` keeps that constructor and the handful of methods used around the path
(``transpose``, ``values``, ``sizes``, ``rename``, ``sortby``, ``isel``, arithmetic for
``preprocess`` callables).  A real ``xr.DataArray`` is accepted wherever this class is (see
``from_any``), so the engine also works in an environment that has xarray.

``data`` may be a numpy array or a torch tensor (host or HBM); nothing here copies it to
the device — that is ``Dataset.to_device``'s job.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from .cfcalendar import CFTimeIndex
from .timegroups import as_time_index


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        self.data = data
        nd = data.ndim if hasattr(data, "ndim") else np.ndim(data)
        if dims is None:
            dims = [f"dim_{i}" for i in range(nd)]
        self.dims = tuple(dims)
        if len(self.dims) != nd:
            raise ValueError(f"dims {self.dims} do not match data with {nd} dimensions")
        self.coords = {}
        for k, v in (coords or {}).items():
            if isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], str):
                v = v[1]
            if k == "time" or isinstance(v, (pd.DatetimeIndex, CFTimeIndex)):
                self.coords[k] = as_time_index(v)
            else:
                self.coords[k] = np.asarray(v)
        for d, n in zip(self.dims, self.shape):
            if d in self.coords and len(self.coords[d]) != n:
                raise ValueError(f"coordinate {d!r} has length {len(self.coords[d])}, dimension has {n}")
        self.name = name
        self.attrs = dict(attrs or {})

    # ---- basic views ----
    @property
    def shape(self):
        return tuple(int(s) for s in self.data.shape)

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def values(self):
        if _is_torch(self.data):
            return self.data.detach().cpu().numpy()
        return np.asarray(self.data)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        raise TypeError("DataArray supports coordinate lookup by name only; use isel for positional indexing")

    def get_index(self, dim):
        return self.coords[dim]

    def __getattr__(self, item):
        coords = self.__dict__.get("coords", {})
        if item in coords:
            return coords[item]
        raise AttributeError(item)

    def _replace(self, data=None, dims=None, coords=None):
        return DataArray(self.data if data is None else data, self.dims if dims is None else dims,
                         dict(self.coords) if coords is None else coords, self.name, self.attrs)

    def copy(self, deep=True):
        if not deep:
            return self._replace()
        data = self.data.clone() if _is_torch(self.data) else np.array(self.data, copy=True)
        return self._replace(data=data)

    # ---- reshaping ----
    def transpose(self, *dims):
        dims = [d for d in dims if d is not Ellipsis]
        rest = [d for d in self.dims if d not in dims]
        order = [self.dims.index(d) for d in list(dims) + rest]
        data = self.data.permute(*order) if _is_torch(self.data) else np.transpose(self.data, order)
        return self._replace(data=data, dims=[self.dims[i] for i in order])

    def rename(self, mapping=None, **kw):
        if isinstance(mapping, str) or mapping is None and not kw:
            out = self._replace()
            out.name = mapping
            return out
        mapping = dict(mapping or {}, **kw)
        dims = [mapping.get(d, d) for d in self.dims]
        coords = {mapping.get(k, k): v for k, v in self.coords.items()}
        return self._replace(dims=dims, coords=coords)

    def isel(self, **indexers):
        data, dims, coords = self.data, list(self.dims), dict(self.coords)
        for d, idx in indexers.items():
            ax = dims.index(d)
            idx_arr = idx if isinstance(idx, slice) else np.asarray(idx)
            sl = [slice(None)] * len(dims)
            sl[ax] = idx_arr if not _is_torch(data) or isinstance(idx_arr, slice) else idx_arr.tolist()
            data = data[tuple(sl)]
            if d in coords:
                coords[d] = coords[d][idx_arr]
            if not isinstance(idx_arr, slice) and idx_arr.ndim == 0:
                dims.pop(ax)
                coords.pop(d, None)
        return DataArray(data, dims, coords, self.name, self.attrs)

    def sel(self, indexers=None, **kw):
        """Label-based selection (the part of ``xarray.DataArray.sel`` the reference's callers use): a
        scalar label drops the dimension, a list / array of labels keeps it, a ``slice`` of labels is
        inclusive on both ends; an absent label raises ``KeyError``."""
        import pandas as pd
        out = self
        for d, lab in dict(indexers or {}, **kw).items():
            c = out.coords[d]
            index = c if isinstance(c, (CFTimeIndex, pd.Index)) else pd.Index(np.asarray(c))
            if isinstance(lab, slice):
                if isinstance(index, CFTimeIndex):
                    raise NotImplementedError("slice selection on a CF-calendar index: use Dataset(time_sel=...)")
                lo, hi = index.slice_locs(lab.start, lab.stop)
                out = out.isel(**{d: slice(lo, hi)})
                continue
            scalar = np.ndim(lab) == 0
            labs = [lab] if scalar else list(lab)
            if isinstance(index, CFTimeIndex):
                vals = list(index)
                pos = []
                for x in labs:
                    if x not in vals:
                        raise KeyError(x)
                    pos.append(vals.index(x))
                pos = np.asarray(pos)
            else:
                if isinstance(index, pd.DatetimeIndex):
                    labs = list(pd.DatetimeIndex(labs))
                pos = index.get_indexer(labs)
                if (pos < 0).any():
                    raise KeyError(labs[int(np.nonzero(pos < 0)[0][0])])
            out = out.isel(**{d: pos[0] if scalar else pos})
        return out

    def expand_dims(self, dim, axis=0):
        """A length-1 dimension ``dim`` (re-)inserted at ``axis``; its coordinate is left out."""
        if dim in self.dims:
            return self
        data = self.data.unsqueeze(axis) if _is_torch(self.data) else np.expand_dims(self.data, axis)
        dims = list(self.dims)
        dims.insert(axis, dim)
        return DataArray(data, dims, dict(self.coords), self.name, self.attrs)

    def sortby(self, dim):
        c = self.coords[dim]
        order = c.argsort() if isinstance(c, CFTimeIndex) else np.argsort(np.asarray(c), kind="stable")
        if np.array_equal(order, np.arange(len(order))):
            return self
        return self.isel(**{dim: order})

    def assign_coords(self, coords=None, **kw):
        new = dict(self.coords)
        for k, v in dict(coords or {}, **kw).items():
            if isinstance(v, tuple):
                v = v[1]
            new[k] = as_time_index(v) if k == "time" else np.asarray(v)
        return self._replace(coords=new)

    def astype(self, dtype):
        if _is_torch(self.data):
            import torch
            return self._replace(data=self.data.to({np.dtype("float32"): torch.float32,
                                                    np.dtype("float64"): torch.float64}[np.dtype(dtype)]))
        return self._replace(data=np.asarray(self.data).astype(dtype))

    # ---- element-wise arithmetic (what preprocess callables use, e.g. x - 273.15) ----
    def _bin(self, other, op):
        o = other.data if isinstance(other, DataArray) else other
        return self._replace(data=op(self.data, o))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b)
    def __neg__(self): return self._replace(data=-self.data)

    def __repr__(self):
        return f"<aggfly_amd.DataArray {self.name or ''} {self.sizes} dtype={self.dtype}>"


def from_any(obj) -> DataArray:
    """Accept our DataArray or anything xarray-like (``dims``, ``coords``, ``data``/``values``)."""
    if isinstance(obj, DataArray):
        return obj
    if hasattr(obj, "dims") and hasattr(obj, "coords"):
        coords = {}
        for k in obj.coords:
            c = obj.coords[k]
            if getattr(c, "ndim", 1) != 1:
                continue
            if k == "time" and hasattr(c, "to_index"):
                idx = c.to_index()
                coords[k] = idx if isinstance(idx, pd.DatetimeIndex) else as_time_index(np.asarray(idx, dtype=object))
            else:
                coords[k] = np.asarray(c.values)
        data = obj.data if isinstance(getattr(obj, "data", None), np.ndarray) else np.asarray(obj.values)
        return DataArray(data, list(obj.dims), coords, getattr(obj, "name", None), dict(getattr(obj, "attrs", {})))
    raise TypeError(f"cannot interpret {type(obj).__name__} as a labelled array")
