"""Generates tests/golden/hdf5/*.nc|*.h5: small HDF5 files written by the REAL HDF5 library (h5py 3.3 / libhdf5
1.10.6 under /opt/conda/bin/python3.9, present in the build container only) in the shapes netCDF-4 files have:
chunked + shuffle + deflate variables, dimension scales, new-style (creation-order) groups, string attributes.
The in-tree reader (aggfly_amd/hdf5.py) is checked against them; expected values come from `recipe()`.

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "hdf5")
T, NY, NX = 37, 9, 14


def recipe():
    rng = np.random.default_rng(2024)
    t2m = (280 + 10 * np.sin(np.arange(T)[:, None, None] / 5.0) + rng.normal(0, 2, (T, NY, NX))).astype("<f4")
    t2m[3, 2, 5] = np.nan
    packed = np.round((np.nan_to_num(t2m, nan=281.3) - 281.3) / 0.0017).astype("<i2")
    packed[np.isnan(t2m)] = -32767
    time = np.arange(T, dtype="<i4") * 6 + 876576          # hours since 1900-01-01 -> 2000-01-01 00:00 onwards, 6-hourly
    lat = (50 - 0.25 * np.arange(NY)).astype("<f4")       # descending, like ERA5
    lon = (230 + 0.25 * np.arange(NX)).astype("<f4")
    return dict(t2m=t2m, packed=packed, time=time, latitude=lat, longitude=lon)


def write(path, h5py, track_order, libver, big_endian=False, fletcher=False):
    r = recipe()
    kw = {} if libver is None else {"libver": libver}
    with h5py.File(path, "w", track_order=track_order, **kw) as f:
        f.attrs["Conventions"] = np.bytes_("CF-1.6")
        f.attrs["history"] = "made by h5py for the aggfly_amd HDF5 reader tests"          # variable-length string
        tm = f.create_dataset("time", data=r["time"], track_order=track_order)
        tm.attrs["units"] = np.bytes_("hours since 1900-01-01 00:00:00.0")
        tm.attrs["calendar"] = np.bytes_("gregorian")
        la = f.create_dataset("latitude", data=r["latitude"].astype(">f4") if big_endian else r["latitude"], track_order=track_order)
        la.attrs["units"] = "degrees_north"                                                  # vlen string attribute
        lo = f.create_dataset("longitude", data=r["longitude"], track_order=track_order)
        lo.attrs["units"] = np.bytes_("degrees_east")
        for d in (tm, la, lo):
            d.make_scale(d.name.strip("/"))
        v = f.create_dataset("t2m", data=r["t2m"], chunks=(5, 4, 6), compression="gzip", compression_opts=4, shuffle=True,
                             fletcher32=fletcher, fillvalue=np.float32(np.nan), track_order=track_order)
        v.attrs["units"] = np.bytes_("K")
        v.attrs["long_name"] = np.bytes_("2 metre temperature")
        v.attrs["_FillValue"] = np.array([np.nan], dtype="<f4")
        p = f.create_dataset("t2m_packed", data=r["packed"], track_order=track_order)                 # contiguous int16
        p.attrs["scale_factor"] = np.array([0.0017], dtype="<f8")
        p.attrs["add_offset"] = np.array([281.3], dtype="<f8")
        p.attrs["_FillValue"] = np.array([-32767], dtype="<i2")
        p.attrs["missing_value"] = np.array([-32767], dtype="<i2")
        c = f.create_dataset("t2m_chunked_nofilter", data=r["t2m"], chunks=(37, 3, 14), track_order=track_order)
        for var in (v, p, c):
            for i, d in enumerate((tm, la, lo)):
                var.dims[i].attach_scale(d)
        g = f.create_group("forecast", track_order=track_order)                                        # a sub-group
        g.create_dataset("lead", data=np.arange(4, dtype="<i8"))


def main():
    import h5py
    os.makedirs(OUT, exist_ok=True)
    write(os.path.join(OUT, "nc4_like.nc"), h5py, track_order=True, libver=None)                      # creation-order groups (netCDF-4 style)
    write(os.path.join(OUT, "old_style.h5"), h5py, track_order=False, libver=None, big_endian=True, fletcher=True)
    write(os.path.join(OUT, "v18.h5"), h5py, track_order=True, libver=("earliest", "v108"))
    write(os.path.join(OUT, "latest.h5"), h5py, track_order=True, libver="latest")
    with h5py.File(os.path.join(OUT, "unlimited_time.nc"), "w", track_order=True) as f:               # a record dimension, grown in steps
        r = recipe()
        tm = f.create_dataset("time", shape=(0,), maxshape=(None,), dtype="<i4", chunks=(8,))
        la = f.create_dataset("latitude", data=r["latitude"])
        lo = f.create_dataset("longitude", data=r["longitude"])
        v = f.create_dataset("t2m", shape=(0, NY, NX), maxshape=(None, NY, NX), dtype="<f4", chunks=(1, NY, NX), compression="gzip",
                             compression_opts=2, shuffle=True, fillvalue=np.float32(9.96921e36))
        tm.attrs["units"] = np.bytes_("hours since 1900-01-01 00:00:00.0")
        for d, n in ((tm, "time"), (la, "latitude"), (lo, "longitude")):
            d.make_scale(n)
        for i, d in enumerate((tm, la, lo)):
            v.dims[i].attach_scale(d)
        for k0 in range(0, T, 10):                                                                     # appended like a model writes records
            k1 = min(T, k0 + 10)
            tm.resize((k1,)); v.resize((k1, NY, NX))
            tm[k0:k1] = r["time"][k0:k1]
            v[k0:k1] = r["t2m"][k0:k1]
    with h5py.File(os.path.join(OUT, "latest_indices.h5"), "w", libver="latest") as f:                # version-4 layouts, every index type
        a = np.arange(42, dtype="<f4").reshape(6, 7)
        f.create_dataset("single", data=a, chunks=(6, 7))
        f.create_dataset("single_filtered", data=a, chunks=(6, 7), compression="gzip", shuffle=True)
        b = np.arange(1200, dtype="<f4").reshape(40, 30)
        f.create_dataset("paged", data=b, chunks=(1, 1), compression="gzip")                          # 1,200 chunks: a paged fixed array
        f.create_dataset("growing", data=a, chunks=(2, 7), maxshape=(None, 7))                        # extensible array: refused
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_chunk((2, 7)); dcpl.set_alloc_time(h5py.h5d.ALLOC_TIME_EARLY)
        space = h5py.h5s.create_simple((6, 7))
        did = h5py.h5d.create(f.id, b"implicit", h5py.h5t.IEEE_F32LE, space, dcpl=dcpl)               # early allocation, no filter: implicit index
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, a)
    with h5py.File(os.path.join(OUT, "dense_group.h5"), "w", track_order=True) as f:                   # > 8 links: fractal heap
        for i in range(12):
            f.create_dataset(f"v{i:02d}", data=np.arange(3, dtype="<f4") + i)
    with h5py.File(os.path.join(OUT, "dense_big.h5"), "w", track_order=True) as f:                     # heap with an indirect root block
        for i in range(150):
            d = f.create_dataset(f"variable_with_a_long_name_{i:03d}", data=np.arange(4, dtype="<i2") + i)
        v = f.create_dataset("t2m", data=np.arange(6, dtype="<f4").reshape(2, 3), track_order=True)
        for k in range(14):                                                                            # > 8 attributes: dense attribute storage
            v.attrs[f"attr_{k:02d}"] = np.array([k * 1.5], dtype="<f8")
        v.attrs["units"] = np.bytes_("K")
        v.attrs["comment"] = "variable-length text among dense attributes"
    with h5py.File(os.path.join(OUT, "many_old_style.h5"), "w") as f:                                  # symbol-table group, several B-tree leaves
        for i in range(40):
            f.create_dataset(f"var_{i:03d}", data=np.arange(5, dtype="<i4") * i)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes", "hdf5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
