"""The built-in HDF5 / netCDF-4 reader (aggfly_amd/hdf5.py) against files written by the REAL HDF5 library
(h5py 3.3 / libhdf5 1.10.6; tests/golden/hdf5/, made by tests/golden/make_hdf5_fixtures.py).  CPU only."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_hdf5_fixtures import recipe          # noqa: E402

import aggfly_amd as af                        # noqa: E402
from aggfly_amd import hdf5                    # noqa: E402

FIX = os.path.join(HERE, "golden", "hdf5")
R = recipe()


@pytest.mark.parametrize("fn", ["nc4_like.nc", "old_style.h5", "v18.h5"])
def test_netcdf4_shaped_files(fn):
    """Creation-order groups (netCDF-4 style), symbol-table groups, v1.8 bounds: chunked + shuffle + deflate
    (+ fletcher32), contiguous int16, big-endian floats, dimension scales, fixed and variable-length strings."""
    with hdf5.H5File(os.path.join(FIX, fn)) as f:
        assert set(f.datasets) == {"time", "latitude", "longitude", "t2m", "t2m_packed", "t2m_chunked_nofilter", "forecast/lead"}
        assert f.attrs["Conventions"] == "CF-1.6" and f.attrs["history"].startswith("made by h5py")
        for name, key in (("t2m", "t2m"), ("t2m_packed", "packed"), ("time", "time"), ("latitude", "latitude"),
                          ("longitude", "longitude"), ("t2m_chunked_nofilter", "t2m")):
            ds = f.datasets[name]
            got = ds.read(threads=3)
            assert got.dtype == R[key].dtype.newbyteorder("=") and got.dtype.isnative
            np.testing.assert_array_equal(got, R[key])
        t2m = f.datasets["t2m"]
        assert t2m.dims == ("time", "latitude", "longitude") and t2m.layout[0] == "chunked" and t2m.layout[2] == (5, 4, 6)
        assert [fid for fid, _ in t2m.filters][:2] == [2, 1]                           # shuffle, then deflate
        assert t2m.attrs["units"] == "K" and t2m.attrs["long_name"] == "2 metre temperature" and np.isnan(t2m.attrs["_FillValue"])
        assert f.datasets["latitude"].attrs["units"] == "degrees_north"                 # a variable-length string
        assert f.datasets["time"].attrs["units"] == "hours since 1900-01-01 00:00:00.0"
        p = f.datasets["t2m_packed"]
        assert p.attrs["scale_factor"] == 0.0017 and p.attrs["add_offset"] == 281.3 and p.attrs["_FillValue"] == -32767
        np.testing.assert_array_equal(f.datasets["forecast/lead"].read(), np.arange(4))
        if fn == "old_style.h5":
            assert f.datasets["latitude"].disk_dtype == np.dtype(">f4") and 3 in [fid for fid, _ in t2m.filters]


def test_dense_groups_and_attributes():
    """More than 8 links / attributes: stored in fractal heaps, found through the v2 B-tree name index (h5py
    renames attributes through temporaries, so the heap has holes: a front-to-back walk would miss entries)."""
    with hdf5.H5File(os.path.join(FIX, "dense_group.h5")) as f:
        assert sorted(f.datasets) == [f"v{i:02d}" for i in range(12)]
        assert all(np.array_equal(f.datasets[f"v{i:02d}"].read(), np.arange(3, dtype="f4") + i) for i in range(12))
    with hdf5.H5File(os.path.join(FIX, "dense_big.h5")) as f:                          # heap with an indirect root, B-tree of depth 1
        assert len(f.datasets) == 151
        assert all(np.array_equal(f.datasets[f"variable_with_a_long_name_{i:03d}"].read(), np.arange(4) + i) for i in range(150))
        a = f.datasets["t2m"].attrs
        assert len(a) == 16 and a["units"] == "K" and a["comment"].startswith("variable-length text")
        assert [a[f"attr_{k:02d}"] for k in range(14)] == [1.5 * k for k in range(14)]
    with hdf5.H5File(os.path.join(FIX, "many_old_style.h5")) as f:                     # symbol table over several B-tree leaves
        assert len(f.datasets) == 40 and all(np.array_equal(f.datasets[f"var_{i:03d}"].read(), np.arange(5) * i) for i in range(40))


def test_record_dimension_grown_in_steps():
    """An unlimited time dimension written record block by record block (v1 B-tree with many small chunks)."""
    with hdf5.H5File(os.path.join(FIX, "unlimited_time.nc")) as f:
        d = f.datasets["t2m"]
        assert d.shape == (37, 9, 14) and d.layout[2] == (1, 9, 14) and d.dims == ("time", "latitude", "longitude")
        np.testing.assert_array_equal(d.read(), R["t2m"])
        np.testing.assert_array_equal(f.datasets["time"].read(), R["time"])
    ds = af.dataset_from_path(os.path.join(FIX, "unlimited_time.nc"), "t2m")
    assert ds.cube().shape == (37, 9, 14) and ds.time[0] == pd.Timestamp("2000-01-01")


def test_latest_format_bounds():
    """Version-4 layouts (HDF5 >= 1.10, "latest" bounds; not what netCDF-4 writes): single-chunk, implicit and
    fixed-array (also paged) chunk indices are read, the indices of growing datasets are refused clearly."""
    with hdf5.H5File(os.path.join(FIX, "latest.h5")) as f:
        for name, key in (("t2m", "t2m"), ("t2m_chunked_nofilter", "t2m"), ("t2m_packed", "packed")):
            np.testing.assert_array_equal(f.datasets[name].read(), R[key])
        assert f.datasets["t2m"].dims == ("time", "latitude", "longitude")
    a, b = np.arange(42, dtype="f4").reshape(6, 7), np.arange(1200, dtype="f4").reshape(40, 30)
    with hdf5.H5File(os.path.join(FIX, "latest_indices.h5")) as f:
        for name, want in (("single", a), ("single_filtered", a), ("paged", b), ("implicit", a)):
            np.testing.assert_array_equal(f.datasets[name].read(), want)
        with pytest.raises(hdf5.HDF5Error, match="growing"):
            f.datasets["growing"].read()
    assert not hdf5.is_hdf5(os.path.join(HERE, "golden", "reference_goldens.json"))
    with pytest.raises(hdf5.HDF5Error):
        hdf5.H5File(os.path.join(HERE, "golden", "reference_goldens.json"))


def test_dataset_from_path_opens_netcdf4():
    """`.nc` inputs as the reference takes them (`dataset.py:636-740`): variable, CF decoding, coordinates, time."""
    path = os.path.join(FIX, "nc4_like.nc")
    ds = af.dataset_from_path(path, "t2m", preprocess=lambda x: x - 273.15)
    np.testing.assert_array_equal(ds.cube(), R["t2m"] - np.float32(273.15))
    assert ds.time.equals(pd.date_range("2000-01-01", periods=37, freq="6h"))
    np.testing.assert_array_equal(ds.latitude, R["latitude"])
    np.testing.assert_array_equal(ds.longitude, R["longitude"])
    packed = af.dataset_from_path(path, "t2m_packed")                                     # int16 + scale / offset / fill
    want = np.where(R["packed"] == -32767, np.nan, R["packed"].astype(np.float32) * np.float32(0.0017) + np.float32(281.3))
    np.testing.assert_array_equal(packed.cube(), want)
    assert packed.cube().dtype == np.float32 and np.isnan(packed.cube()).sum() == 1
    sel = af.dataset_from_path(path, "t2m", time_sel=slice("2000-01-03", "2000-01-04"))
    assert len(sel.time) == 8
    with pytest.raises(KeyError):
        af.dataset_from_path(path, "no_such_variable")
