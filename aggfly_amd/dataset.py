"""``Dataset`` / ``Grid``: the host-side input wrapper of the hot path.

Mirrors `aggfly/dataset/dataset.py:21-130,419-563` and `aggfly/dataset/grid.py:19-217` as
far as the aggregation path reads them: dimension normalisation to
``(latitude, longitude, time)`` (`aggfly/dataset/grid_utils.py:299-324`), time sort
(`dataset.py:88`), ``time_sel``, lazy-in-the-reference / eager-here ``preprocess``
(`dataset.py:94-95`), the 0-360 <-> +-180 longitude re-sort (`dataset.py:419-440`,
`grid_utils.py:16-73`) and the positional ``cell_id`` contract (`grid.py:137-147,214-217`).

What differs by design: there is no dask graph.  ``Dataset.da`` wraps a plain array (numpy
on the host, or a torch tensor already resident in HBM after ``to_device()``); the engine
consumes the time-major ``(time, lat, lon)`` cube directly (``Dataset.cube()``), which is
the layout climate stores have on disk.  ``power`` / ``interact`` / ``spline`` run the library's
element-wise kernel (`afhip_transform`) on the current array in HBM (a host array is uploaded first;
without a GPU they raise); inside ``aggregate_dataset`` those transforms are fused into the streaming
kernel instead of materialising new datasets.
"""
from __future__ import annotations

import copy

import numpy as np
import pandas as pd

from .dataarray import DataArray, _is_torch, from_any


def lon_to_180(lon):
    """`grid_utils.py:16-31`."""
    return (np.asarray(lon, dtype=float) + 180) % 360 - 180


def lon_to_360(lon):
    """`grid_utils.py:34-49`."""
    lon = np.asarray(lon, dtype=float)
    return (lon < 0) * (lon + 360) + (lon >= 0) * lon


def clean_dims(da: DataArray, xycoords, timecoord) -> DataArray:
    """`grid_utils.py:299-324`: canonical names, then (latitude, longitude, time, ...)."""
    if tuple(xycoords) != ("longitude", "latitude"):
        da = da.rename({xycoords[0]: "longitude", xycoords[1]: "latitude"})
    if timecoord != "time":
        da = da.rename({timecoord: "time"})
    return da.transpose("latitude", "longitude", "time", ...)


class Grid:
    """Positional grid metadata (`grid.py:19-217`).  ``cell_id`` is the row-major
    (latitude, longitude) position on THIS grid; clipping re-indexes it from 0."""

    def __init__(self, longitude, latitude, name=None, lon_is_360=True):
        self.longitude = np.asarray(longitude, dtype=float)
        self.latitude = np.asarray(latitude, dtype=float)
        self.name = name
        self.lon_is_360 = lon_is_360
        self._reindex()
        self.resolution_lon, self.resolution_lat = self.get_resolution()

    def _reindex(self):
        self.lon_array, self.lat_array = np.meshgrid(self.longitude, self.latitude)
        self.index = np.arange(self.lon_array.size).reshape(self.lon_array.shape)
        self.cell_id = self.index.flatten()
        self._cell_key = (id(self.cell_id), "arange", int(self.cell_id.size))     # cheap cache key while cell_id is this very array

    @property
    def resolution(self):
        return max(self.resolution_lon, self.resolution_lat)

    @property
    def is_square(self):
        return bool(np.isclose(self.resolution_lon, self.resolution_lat))

    def get_resolution(self):
        res_lon = abs(np.diff(self.longitude).mean()) if len(self.longitude) > 1 else 0.0
        res_lat = abs(np.diff(self.latitude).mean()) if len(self.latitude) > 1 else 0.0
        if res_lon == 0.0:
            res_lon = res_lat
        if res_lat == 0.0:
            res_lat = res_lon
        return res_lon, res_lat

    def clip_grid_to_bbox(self, bounds):
        """`grid.py:176-217`: keep centroids within half a cell of [minx, miny, maxx, maxy]."""
        inlon = (self.longitude >= bounds[0] - self.resolution_lon / 2) & (self.longitude <= bounds[2] + self.resolution_lon / 2)
        inlat = (self.latitude >= bounds[1] - self.resolution_lat / 2) & (self.latitude <= bounds[3] + self.resolution_lat / 2)
        self.longitude = self.longitude[inlon]
        self.latitude = self.latitude[inlat]
        self._reindex()
        return inlat, inlon

    def clip_grid_to_georegions_extent(self, georegions):
        """`grid.py:150-174`."""
        bounds = np.array(georegions.total_bounds, dtype=float)
        if self.lon_is_360:
            allb = lon_to_360(np.asarray(georegions.bounds)[:, [0, 2]])
            bounds[[0, 2]] = [allb[:, 0].min(), allb[:, 1].max()]
        return self.clip_grid_to_bbox(bounds)


class Dataset:
    """A climate variable on a regular grid (`dataset.py:21-130`)."""

    def __init__(self, da, xycoords=("longitude", "latitude"), timecoord="time", time_sel=None,
                 lon_is_360=True, preprocess=None, georegions=None, time_fix=False, name=None):
        da = clean_dims(from_any(da), xycoords, timecoord)
        if "time" in da.coords:
            da = da.sortby("time")
        if time_sel is not None:
            da = self._time_sel(da, time_sel)
        if preprocess is not None:
            da = preprocess(da)
        self.da = da
        self.name = name
        self.lon_is_360 = lon_is_360
        assert all(k in self.da.coords for k in ("latitude", "longitude"))
        self.grid = Grid(self.longitude, self.latitude, self.name, self.lon_is_360)
        self.history = []
        self.georegions = georegions
        if georegions is not None:
            self.clip_data_to_georegions_extent(georegions)
        if time_fix:
            raise NotImplementedError("time_fix is deprecated in the reference (dataset.py:55) and not provided here")

    # ---- coordinates ----
    @property
    def coords(self):
        return self.da.coords

    @property
    def longitude(self):
        return np.asarray(self.da.coords["longitude"])

    @property
    def latitude(self):
        return np.asarray(self.da.coords["latitude"])

    @property
    def time(self):
        return self.da.coords["time"]

    @staticmethod
    def _time_sel(da, time_sel):
        t = da.coords["time"]
        if isinstance(t, pd.DatetimeIndex):
            loc = t.slice_indexer(time_sel.start, time_sel.stop) if isinstance(time_sel, slice) else t.get_loc(time_sel)
            if isinstance(loc, (int, np.integer)):
                loc = slice(loc, loc + 1)
            return da.isel(time=loc if isinstance(loc, slice) else np.asarray(loc))
        return da.isel(time=t.sel_positions(time_sel))       # CF calendar: year / month / day granular, like xarray

    # ---- the engine's view ----
    def cube(self):
        """(time, latitude, longitude) array, C-contiguous — what the kernels stream.

        ``da`` is normally a permuted view of exactly that layout (stores are time-major), so
        this is free; otherwise one copy is made."""
        d = self.da.transpose("time", "latitude", "longitude").data
        if _is_torch(d):
            return d.contiguous()
        return np.ascontiguousarray(d)

    def to_device(self, device="cuda"):
        """Move the cube into HBM once (float32 / float64 kept as stored)."""
        import torch
        d = self.cube()
        if not _is_torch(d):
            d = torch.from_numpy(d)      # a plain pageable copy already runs at PCIe rate here (~56 GB/s measured)
        d = d.to(device, non_blocking=True)
        tm = DataArray(d, ("time", "latitude", "longitude"),
                       {k: self.da.coords[k] for k in ("time", "latitude", "longitude")}, self.da.name, self.da.attrs)
        self.da = tm.transpose("latitude", "longitude", "time")
        return self

    # ---- reference API ----
    def deepcopy(self):
        new = copy.copy(self)
        new.da = self.da.copy(deep=False)       # arrays are immutable inputs to the engine
        new.grid = copy.deepcopy(self.grid)
        new.history = list(self.history)
        return new

    def update(self, array, **_):
        """`dataset.py:225-297` for the forms the path uses: a labelled array replaces
        ``da``; a bare array replaces the data and must keep the shape."""
        if isinstance(array, DataArray) or (hasattr(array, "dims") and hasattr(array, "coords")):
            self.da = from_any(array)
        else:
            if tuple(array.shape) != self.da.shape:
                raise ValueError("update(): bare arrays must keep the current shape")
            self.da = self.da._replace(data=array)

    def sel(self, **kwargs):
        """`dataset.py:401-417`: select by label along the named dimensions, keeping every dimension
        (a scalar label leaves a length-1 axis, as the reference's ``expand_dims`` does)."""
        da = self.da
        for k, v in kwargs.items():
            da = da.sel(**{k: [v] if np.ndim(v) == 0 and not isinstance(v, slice) else v})
        self.da = da
        self.grid = Grid(self.longitude, self.latitude, self.name, self.lon_is_360)

    def rechunk(self, chunks="auto"):
        """`dataset.py:120-130`.  There is no chunk graph here (the cube is one resident array): accepted, no effect."""
        return None

    def compute(self, dask_array=True, chunks=None):
        """`dataset.py:196-211`.  Data are always materialised: accepted, no effect."""
        return None

    def interior_cells(self, georegions, buffer=None, dtype="georegions", maxsize=None):
        """`dataset.py:315-399` builds cell polygons with geopandas / shapely — part of the CPU-side weights
        pipeline that stays with the reference."""
        raise ImportError("Dataset.interior_cells belongs to aggfly's CPU-side geometry pipeline (geopandas / shapely), "
                          "which this engine does not re-implement; run it with aggfly and pass the table to weights_from_objects(table=...)")

    def clip_data_to_grid(self, inlat, inlon):
        self.da = self.da.isel(latitude=np.nonzero(inlat)[0], longitude=np.nonzero(inlon)[0])

    def clip_data_to_georegions_extent(self, georegions, update=True):
        """`dataset.py:150-175`."""
        tgt = self if update else self.deepcopy()
        inlat, inlon = tgt.grid.clip_grid_to_georegions_extent(georegions)
        tgt.clip_data_to_grid(inlat, inlon)
        return None if update else tgt

    def clip_data_to_bbox(self, bounds):
        inlat, inlon = self.grid.clip_grid_to_bbox(bounds)
        self.clip_data_to_grid(inlat, inlon)

    def rescale_longitude(self):
        """`dataset.py:419-440`: flip between 0-360 and +-180 and re-sort by longitude."""
        lon = lon_to_180(self.longitude) if self.lon_is_360 else lon_to_360(self.longitude)
        self.da = self.da.assign_coords(longitude=lon).sortby("longitude")
        self.lon_is_360 = not self.lon_is_360
        self.grid = Grid(self.longitude, self.latitude, self.name, self.lon_is_360)

    def lon_order_to_180(self):
        """Column order that ``rescale_longitude`` would apply to a 0-360 grid, without
        touching the data: the engine folds this permutation into the CSR columns."""
        if not self.lon_is_360:
            return np.arange(len(self.longitude)), self.longitude
        lon = lon_to_180(self.longitude)
        order = np.argsort(lon, kind="stable")
        return order, lon[order]

    def _with(self, data, tag):
        new = self.deepcopy()
        new.da = self.da._replace(data=data)
        new.history.append(tag)
        return new

    # ---- element-wise transforms: `k_transform` of the HIP library (include/aggfly_hip.h: afhip_transform) ----
    def _hbm_time_major(self):
        """The (time, latitude, longitude) cube as a float HBM tensor, uploaded if still on the host.  Raises
        `HipEngineError` without a GPU: the transforms have no CPU form here."""
        import torch
        from . import hip
        hip.require_gpu()
        d = self.cube()
        if not _is_torch(d):
            if d.dtype not in (np.float32, np.float64):
                d = d.astype(np.float64)
            d = torch.from_numpy(d)
        if not d.is_cuda:
            d = d.cuda(non_blocking=True)
        if d.dtype not in (torch.float32, torch.float64):
            d = d.to(torch.float64)
        return d.contiguous()

    def _transformed(self, tm, tag, update):
        """Wrap a time-major HBM result as this dataset's ``da`` (dimension order kept: a permuted view)."""
        da = DataArray(tm, ("time", "latitude", "longitude"),
                       {k: self.da.coords[k] for k in ("time", "latitude", "longitude")}, self.da.name, self.da.attrs)
        da = da.transpose(*self.da.dims) if set(self.da.dims) == {"time", "latitude", "longitude"} else da
        if update:
            self.da = da
            self.history.append(tag)
            return None
        new = self.deepcopy()
        new.da = da
        new.history.append(tag)
        return new

    def power(self, exp, update=False):
        """`dataset.py:442-473` (`np.power` per block, `:527-543`), on the GPU: integer exponents through the kernel's
        correctly rounded product chain, others through ``pow``."""
        import torch
        from . import hip
        x = self._hbm_time_major()
        # NumPy 2 promotion of np.power(block, exp): a Python scalar is weak (float32 stays float32); a NumPy scalar brings its
        # dtype (np.int32 / np.int64 / np.float64 promote float32 to float64, np.int16 / np.float32 do not)
        out_dtype = x.dtype
        if isinstance(exp, np.generic):
            promoted = np.result_type(np.float32 if x.dtype == torch.float32 else np.float64, exp.dtype)
            out_dtype = torch.float64 if promoted == np.float64 else x.dtype
        out = hip.transform(x, "pow", float(exp), out_dtype=out_dtype)
        return self._transformed(out, f"power{exp}", update)

    def spline(self):
        """`dataset.py:475-481`: (self, hinge at 20)."""
        from . import hip
        return self, self._transformed(hip.transform(self._hbm_time_major(), "hinge", 20.0), "spline", False)

    def interact(self, inter, update=False):
        """`dataset.py:483-518`: element-wise product with a second array of the same shape (`_interact`, `:547-563`)."""
        import torch
        from . import hip
        if isinstance(inter, Dataset):
            inter = inter.da
        if isinstance(inter, DataArray):
            if set(inter.dims) == set(self.da.dims):
                inter = inter.transpose(*self.da.dims)      # same layout as self before the shape check
            other = inter.data
        else:
            other = inter
        assert tuple(self.da.data.shape) == tuple(other.shape)
        x = self._hbm_time_major()
        # the second array in the cube's own (time, latitude, longitude) order: layout only, the product runs in the kernel
        perm = [self.da.dims.index(d) for d in ("time", "latitude", "longitude")]
        if not _is_torch(other):
            other = np.asarray(other)
            if other.dtype not in (np.float32, np.float64):
                other = other.astype(np.float64)
            other = torch.from_numpy(np.ascontiguousarray(other))
        other = other.to(x.device, non_blocking=True)
        if other.dtype not in (torch.float32, torch.float64):
            other = other.to(torch.float64)
        other = other.permute(*perm).contiguous()
        out_dtype = torch.float32 if (x.dtype == torch.float32 and other.dtype == torch.float32) else torch.float64
        out = hip.transform(x, "inter", other=other, out_dtype=out_dtype)
        return self._transformed(out, "interacted", update)

    def __repr__(self):
        return f"<aggfly_amd.Dataset {self.name or ''} {self.da.sizes} lon_is_360={self.lon_is_360}>"
