"""Host logic that needs no GPU: spec lowering, group bounds, calendars, CSR triplets,
the Zarr codec, the C-ABI surface.  CPU only."""
import ctypes
import os
import re
import sys
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_inputs as gi  # noqa: E402

import aggfly_amd as af
from aggfly_amd import cfcalendar as cfc, engine as eng, hip, timegroups as tg
from oracle import ref_temporal as rt
from oracle.ref_calendar import cf_daily_index
from oracle.ref_spatial import weight_triplets as ref_triplets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = hip.load()
    hdr = open(os.path.join(ROOT, "include", "aggfly_hip.h")).read()
    declared = set(re.findall(r"\b(afhip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.afhip_abi_version() == hip.ABI_VERSION == 4
    assert int(re.search(r"#define AFHIP_ABI_VERSION (\d+)", hdr).group(1)) == hip.ABI_VERSION
    assert isinstance(hip.device_count(), int)


def test_struct_layout_matches_header():
    assert ctypes.sizeof(hip.Column) == 4 * 4 + 8 * 7
    assert ctypes.sizeof(hip.PlanDesc) == 8 + 8 + 4 + 4 + 8 + 8 + 8 + 8 + 8 + 4 + 4


def test_product_fails_loudly_without_gpu():
    if hip.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(hip.HipEngineError, match="no HIP device"):
        hip.require_gpu()
    arr, time, lat, lon = gi.dataset_360_inputs()
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}))
    with pytest.raises(hip.HipEngineError):
        af.aggregate_time(ds, tavg=[("aggregate", {"calc": "mean", "groupby": "date"})])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "aggfly_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn


def test_lower_spec_key_naming_and_fanout():
    spec = gi.g1_spec()
    keys = []
    for name, steps in spec.items():
        cols, fusable = eng.lower_spec(name, steps)
        assert fusable
        keys += [c.key for c in cols]
    assert keys == gi.goldens()["G1_temporal_table"]["columns"]
    cols, fus = eng.lower_spec("x", [("aggregate", {"calc": "mean", "groupby": "date"}),
                                     ("transform", {"transform": "spline"}),
                                     ("aggregate", {"calc": "sum", "groupby": "year"})])
    assert fus and [c.key for c in cols] == ["x_spline1", "x_spline2"] and cols[0].tf is None and cols[1].tf == ("hinge", 20.0)
    # three aggregate levels / transform on raw data are valid but run staged
    assert not eng.lower_spec("x", [("aggregate", {"calc": "mean", "groupby": "date"}),
                                    ("aggregate", {"calc": "mean", "groupby": "month"}),
                                    ("aggregate", {"calc": "sum", "groupby": "year"})])[1]
    assert not eng.lower_spec("x", [("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                                    ("aggregate", {"calc": "mean", "groupby": "date"})])[1]


def test_lower_spec_errors_match_reference():
    with pytest.raises(ValueError, match="multiple ddargs"):
        eng.lower_spec("x", [("aggregate", {"calc": "mean", "groupby": "date"}),
                             ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                             ("aggregate", {"calc": "bins", "groupby": "month", "ddargs": [[0, 1, 0], [1, 2, 0]]})])
    with pytest.raises(KeyError):
        eng.lower_spec("x", [("aggregate", {"calc": "mean", "groupby": "decade"})])
    with pytest.raises(ValueError, match="No valid transform"):
        eng.lower_spec("x", [("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "log"})])
    with pytest.raises(ValueError, match="engine must be"):
        af.resolve_engine("bogus")
    assert af.resolve_engine("auto") == "hip" and af.resolve_engine("hip") == "hip"


@pytest.mark.parametrize("freq", ["1D", "ME", "YE", "W"])
def test_resample_groups_datetime_matches_oracle(freq):
    t = pd.date_range("2000-12-30 06:00", periods=24 * 40, freq="6h")
    t = t.delete(slice(100, 230))                       # a gap spanning whole days
    b, lab = tg.resample_groups(t, freq)
    bo, labo = rt.resample_groups(t, freq)
    assert b.tolist() == bo.tolist() and lab.equals(labo)
    with pytest.raises(ValueError, match="monotonic"):
        tg.resample_groups(t[::-1], freq)


@pytest.mark.parametrize("calendar", ["360_day", "noleap", "all_leap"])
@pytest.mark.parametrize("freq", ["1D", "ME", "YE"])
def test_cf_calendar_groups_match_independent_oracle(calendar, freq):
    n = 800
    ti = af.cf_range("2000-01-01", n, "D", calendar)
    to = cf_daily_index(calendar, n)
    keep = np.r_[0:40, 75:400, 401:n]                   # gaps incl. a whole missing month
    b, lab = tg.resample_groups(ti[keep], freq)
    bo, labo = rt.resample_groups(to[keep], freq)
    assert b.tolist() == bo.tolist()
    assert len(lab) == len(labo)
    for a, o in zip(lab, labo):
        assert (a.year, a.month, a.day, a.hour) == (o.year, o.month, o.day, o.hour) and a.calendar == calendar


def test_cf_calendar_basics_K2():
    k = gi.goldens()["K2_bounds"]
    t360 = af.cf_range("2000-01-01", 720, "D", "360_day")
    b_m, lab_m = tg.resample_groups(t360, "ME")
    assert set(np.diff(b_m).tolist()) == {k["360_day_720_ME_group_size"]} and len(lab_m) == k["360_day_720_ME_n_labels"]
    assert isinstance(lab_m, af.CFTimeIndex) and str(lab_m[1]) == "2000-02-30 00:00:00"
    assert tg.resample_groups(t360, "YE")[0].tolist() == k["360_day_720_YE_bounds"]
    assert np.diff(tg.resample_groups(af.cf_range("2000-01-01", 365, "D", "noleap"), "ME")[0])[:3].tolist() == k["noleap_365_ME_first3"]
    with pytest.raises(NotImplementedError, match="week"):
        cfc.resample_bins(t360, "W")
    dec = cfc.decode_cf_time([0, 1, 59], "days since 2000-01-01", "noleap")
    assert str(dec[2]) == "2000-03-01 00:00:00"


def test_weight_triplets_match_reference_semantics():
    wdf = pd.DataFrame({"cell_id": [3, 0, 7, 99, 2, 2], "index_right": [5, 5, 2, 2, 9, 2], "weight": [.1, .2, .3, .4, .5, .6]})
    cell_ids = np.arange(8)
    for cids in (cell_ids, np.array([7, 6, 5, 4, 3, 2, 1, 0])):
        r, c, w, ids = eng.weight_triplets(wdf, cids)
        ro, co, wo, idso = ref_triplets(wdf, cids)
        assert r.tolist() == ro.tolist() and c.tolist() == co.tolist() and w.tolist() == wo.tolist() and ids.tolist() == idso.tolist()


def test_dataset_normalisation_and_lon_resort():
    arr, time, lat, lon = gi.dataset_360_inputs()
    da = af.DataArray(data=arr, dims=["time", "latitude", "longitude"], coords={"time": time, "latitude": lat, "longitude": lon})
    ds = af.Dataset(da, lon_is_360=True)
    assert ds.da.dims == ("latitude", "longitude", "time") and ds.cube().shape == (4, 2, 2)
    order, lon180 = ds.lon_order_to_180()
    assert order.tolist() == [1, 0] and lon180.tolist() == [-90.0, 90.0]
    ds2 = ds.deepcopy(); ds2.rescale_longitude()
    assert not ds2.lon_is_360 and ds2.longitude.tolist() == [-90.0, 90.0]
    assert np.array_equal(ds2.cube(), arr[:, :, ::-1])
    assert ds.lon_is_360 and ds.longitude.tolist() == [90.0, 270.0]           # deepcopy left the original alone
    assert ds2.grid.cell_id.tolist() == [0, 1, 2, 3]
    w = af.weights_from_objects(ds, af.GeoRegions(pd.DataFrame({"geoid": ["region_1"]})), table=gi.g2_weights_table())
    assert w.zero_weight == "nan" and not w.grid.lon_is_360
    with pytest.raises(ValueError, match="zero_weight must be one of"):
        af.weights_from_objects(ds, af.GeoRegions(pd.DataFrame({"geoid": ["r"]})), zero_weight="bogus")


def test_dataset_sel_keeps_every_dimension():
    """`Dataset.sel` (`dataset.py:401-417`): label selection that keeps the selected axis; rechunk / compute are
    accepted no-ops; interior_cells points at the CPU-side pipeline."""
    arr, time, lat, lon = gi.dataset_360_inputs()
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}), lon_is_360=True)
    one = ds.deepcopy(); one.sel(time=time[2])
    assert one.da.dims == ("latitude", "longitude", "time") and one.cube().shape == (1, 2, 2)
    assert np.array_equal(one.cube()[0], arr[2]) and one.time.equals(time[2:3])
    two = ds.deepcopy(); two.sel(latitude=[lat[1]], time=[time[0], time[3]])
    assert two.cube().shape == (2, 1, 2) and np.array_equal(two.cube()[:, 0], arr[[0, 3], 1])
    assert two.grid.latitude.tolist() == [lat[1]]
    rng = ds.deepcopy(); rng.sel(time=slice(time[1], time[2]))
    assert rng.time.equals(time[1:3])                                            # label slices are inclusive
    with pytest.raises(KeyError):
        ds.deepcopy().sel(latitude=12.345)
    assert ds.rechunk("auto") is None and ds.compute() is None and ds.cube().shape == (4, 2, 2)
    with pytest.raises(ImportError, match="geopandas"):
        ds.interior_cells(None)


def test_transform_dataset_and_multi_dd_keys():
    """`transform_dataset` / `multi_dd_to_dict` (`aggregate.py:36-78,285-303`): the eager helpers run the library's
    element-wise kernel — without a GPU they fail loudly instead of computing in numpy (their numbers are checked on
    the GPU, `test_gpu_api.py::test_dataset_transforms_run_in_the_hip_library`); the key naming is host logic."""
    from aggfly_amd.aggregate import multi_dd_to_dict, transform_dataset
    from aggfly_amd.hip import HipEngineError
    arr, time, lat, lon = gi.dataset_360_inputs()
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}))
    for kw in (dict(transform="power", exp=[np.arange(1, 4)]), dict(transform="spline"), dict(transform="interact", inter=ds)):
        with pytest.raises(HipEngineError):
            transform_dataset(ds, "tavg", **kw)
    with pytest.raises(ValueError, match="No valid transform"):
        transform_dataset(ds, "t", transform="log")
    assert multi_dd_to_dict(["a", "b"], "dd", [[10, 30, 0], [0, 5, 1]]) == (["a", "b"], ["dd_10_30", "dd_0_5"])
    # the DSL lowering names the same outputs: key_{e}, key_spline1/2, and `inter` keeps its key — all three fused
    from aggfly_amd.engine import lower_spec
    agg = ("aggregate", {"calc": "mean", "groupby": "date"})
    cols, fusable = lower_spec("tavg", [agg, ("transform", {"transform": "power", "exp": [np.arange(1, 4)]})])
    assert [c.key for c in cols] == ["tavg_1", "tavg_2", "tavg_3"] and fusable
    cols, fusable = lower_spec("t", [agg, ("transform", {"transform": "spline"})])
    assert [c.key for c in cols] == ["t_spline1", "t_spline2"] and fusable
    cols, fusable = lower_spec("t", [agg, ("transform", {"transform": "inter", "inter": ds}), ("aggregate", {"calc": "sum", "groupby": "month"})])
    assert [c.key for c in cols] == ["t"] and fusable and cols[0].tf[0] == "inter"
    cols, fusable = lower_spec("t", [("transform", {"transform": "inter", "inter": ds}), agg])          # on raw data: staged
    assert not fusable


def test_lat_and_time_windows_on_the_host_route(tmp_path):
    """`dataset_from_path(lat_window=, time_window=)` without a device: the windows are cut after the host read, on the
    grid the regions' clip left — the same rows and steps the streaming route would have read."""
    from aggfly_amd.io import _band_of_box
    arr, time, lat, lon = gi.dataset_360_inputs()
    big = np.tile(arr, (3, 3, 2))[:10]                                               # (10, 6, 4)
    t = pd.date_range("2000-01-01", periods=10, freq="D")
    la, lo = 10.0 + np.arange(6), 100.0 + np.arange(4)
    ds = af.Dataset(af.DataArray(big, ["time", "latitude", "longitude"], {"time": t, "latitude": la, "longitude": lo}), lon_is_360=True)
    store = str(tmp_path / "w.zarr")
    af.dataset_to_zarr(ds, store, var="v", chunks={"time": 4, "latitude": 3, "longitude": 4})
    band = af.dataset_from_path(store, "v", lon_is_360=True, lat_window=(2, 5), time_window=(3, 9))
    assert band.cube().shape == (6, 3, 4) and np.array_equal(band.cube(), big[3:9, 2:5])
    assert band.latitude.tolist() == la[2:5].tolist() and band.grid.latitude.tolist() == la[2:5].tolist() and band.time.equals(t[3:9])
    # the stored-axes box of a band inside a clip box, either dimension order
    assert _band_of_box(("time", "latitude", "longitude"), (10, 6, 4), ("longitude", "latitude"), (1, 5, 0, 4), (1, 3)) == (2, 4, 0, 4)
    assert _band_of_box(("time", "longitude", "latitude"), (10, 4, 6), ("longitude", "latitude"), None, (2, 6)) == (0, 4, 2, 6)
    with pytest.raises(ValueError, match="outside"):
        _band_of_box(("time", "latitude", "longitude"), (10, 6, 4), ("longitude", "latitude"), (1, 5, 0, 4), (0, 5))


def test_gpu_decode_route_is_chosen_by_request_size(tmp_path, monkeypatch):
    """`io._gpu_decodable`: Blosc-LZ4 stores take the decode-in-HBM route for requests of 96 MB or more (256 MB where the chunks hold whole time steps), always with
    AGGFLY_HIP_GPU_DECODE=1, never with =0; other codecs never do."""
    from aggfly_amd import io as afio
    arr, time, lat, lon = gi.dataset_360_inputs()
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}), lon_is_360=True)
    lz4, raw = str(tmp_path / "a.zarr"), str(tmp_path / "b.zarr")
    af.dataset_to_zarr(ds, lz4, var="v")
    af.dataset_to_zarr(ds, raw, var="v", compress=False)
    monkeypatch.delenv("AGGFLY_HIP_GPU_DECODE", raising=False)
    za = afio.ZarrArray(os.path.join(lz4, "v"))                      # (the default chunks hold whole time steps of the grid)
    assert not afio._gpu_decodable(za, 1 << 20) and not afio._gpu_decodable(za, afio.GPU_DECODE_AUTO_BYTES_WHOLE_ROWS - 1)
    assert afio._gpu_decodable(za, afio.GPU_DECODE_AUTO_BYTES_WHOLE_ROWS)
    tiled = str(tmp_path / "t.zarr")
    af.dataset_to_zarr(ds, tiled, var="v", chunks={"time": 2, "latitude": 1, "longitude": 2})
    zt = afio.ZarrArray(os.path.join(tiled, "v"))
    assert not afio._gpu_decodable(zt, 1 << 20) and afio._gpu_decodable(zt, afio.GPU_DECODE_AUTO_BYTES)
    assert za._blosc_geometry[1] == arr.dtype.itemsize
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "1")
    assert afio._gpu_decodable(za, 0)
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "0")
    assert not afio._gpu_decodable(za, 1 << 40)
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "1")
    assert not afio._gpu_decodable(afio.ZarrArray(os.path.join(raw, "v")), 1 << 40)


def test_preprocess_and_unsorted_time():
    arr, time, lat, lon = gi.dataset_360_inputs()
    perm = [2, 0, 3, 1]
    da = af.DataArray(arr[perm] + 273.15, ["time", "latitude", "longitude"], {"time": time[perm], "latitude": lat, "longitude": lon})
    ds = af.Dataset(da, preprocess=lambda x: x - 273.15)
    assert np.allclose(ds.cube(), arr) and ds.time.equals(time)


def test_zarr_roundtrip_and_auto_chunks(tmp_path):
    from aggfly_amd.io import _auto_chunks, _looks_like_zarr
    c = _auto_chunks({"latitude": 721, "longitude": 1440, "time": 8784}, 4, 256)
    assert c["time"] == -1 and c["latitude"] == c["longitude"] and c["latitude"] >= 32
    c = _auto_chunks({"latitude": 721, "longitude": 1440, "time": 350640}, 4, 256)
    assert 0 < c["time"] < 350640 and c["time"] * c["latitude"] * c["longitude"] * 4 <= 256 * 1024 * 1024
    c = _auto_chunks({"latitude": 2, "longitude": 2, "time": 4}, 8, 256)
    assert c["latitude"] <= 2 and c["longitude"] <= 2
    rng = np.random.default_rng(0)
    for cal in (None, "noleap"):
        T = 50
        time = pd.date_range("2001-01-01", periods=T, freq="D") if cal is None else af.cf_range("2001-01-01", T, "D", cal)
        arr = rng.normal(280, 10, (T, 5, 6)).astype(np.float32)
        ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"],
                                     {"time": time, "latitude": np.arange(5.0), "longitude": np.arange(6.0)}), lon_is_360=False)
        store = str(tmp_path / f"s_{cal}.zarr")
        af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 16, "latitude": 3, "longitude": 4})
        assert _looks_like_zarr(store)
        back = af.dataset_from_path(store, var="t2m", lon_is_360=False, preprocess=lambda x: x - 273.15)
        assert np.array_equal(back.cube(), arr - np.float32(273.15))
        if cal is None:
            assert back.time.equals(time)
        else:
            assert back.time == time and back.time.calendar == "noleap"


def test_duplicate_output_keys_follow_dict_semantics():
    """Two ddargs rows with equal bounds but different flags produce the same key: like the
    reference's dict merge (aggregate.py:160-161) the later column wins at the first one's
    position, and the shadowed column is not computed at all."""
    from aggfly_amd.aggregate import _lower_all
    spec = {"a": [("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [[5.0, 95.0, 0], [1.0, 2.0, 0], [5.0, 95.0, 1]]}),
                  ("aggregate", {"calc": "mean", "groupby": "year"})],
            "b": [("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "year"})],
            "a_1.0_2.0": [("aggregate", {"calc": "max", "groupby": "year"})]}
    order, fused, staged, names = _lower_all(spec)
    assert names == ["a_5.0_95.0", "a_1.0_2.0", "b"] and not staged
    by_key = {c.key: c for c in fused}
    assert len(fused) == 3
    assert by_key["a_5.0_95.0"].inner.ddargs == (5.0, 95.0, 1.0)          # the later row won
    assert by_key["a_1.0_2.0"].inner.calc == "max"                        # the later NAME won that key


def test_frame_assembly_and_region_merge_equal_the_reference_recipe():
    """`_assemble_frame` / `_merge_regions` take shortcuts (rows chosen on the [R, P] panel, lookup instead of a
    generic join); on random panels they must give exactly what the reference's pandas recipe gives
    (`spatial.py:136-154`: full frame, NaN-row policy; `aggregate.py:276-280`: merge with the region table)."""
    from aggfly_amd import aggregate as agg

    def recipe(res, names, region_ids, labels, weights):
        n_regions, n_time = res.shape[1], res.shape[2]
        out = pd.DataFrame({"region_id": np.repeat(region_ids, n_time), "time": np.tile(agg._label_values(labels), n_regions)})
        for k, nm in enumerate(names):
            out[nm] = res[k].reshape(-1)
        if getattr(weights, "zero_weight", "area") == "nan":
            zr = agg._zero_weight_regions(weights.weights)
            keep = out["region_id"].isin(zr) | out[list(names)].notna().all(axis=1)
            return out.loc[keep].reset_index(drop=True)
        return out.dropna(subset=list(names)).reset_index(drop=True)

    rng = np.random.default_rng(0)
    for trial in range(40):
        R, P, K = int(rng.integers(1, 30)), int(rng.integers(1, 12)), int(rng.integers(1, 5))
        res = rng.normal(size=(K, R, P))
        res[rng.random((K, R, P)) < 0.15] = np.nan
        if trial % 3 == 0:
            res[:, rng.integers(0, R), :] = np.nan
        names = [f"c{i}" for i in range(K)]
        index = pd.RangeIndex(R + 3) if trial % 3 else pd.Index(np.sort(rng.choice(100, R + 3, replace=False)))
        if trial % 7 == 0:
            index = pd.Index(rng.permutation(R + 3))                      # not ascending: the merge falls back to pandas
        rid = np.sort(rng.choice(np.asarray(index), R, replace=False))
        if trial % 11 == 0:
            rid[-1] = 9999                                                # a region the table does not hold: dropped by the merge
        labels = pd.date_range("2000-01-31", periods=P, freq="ME") if trial % 2 else af.cf_range("2000-01-01", P, "D", "noleap")
        wdf = pd.DataFrame({"index_right": np.repeat(rid, 2), "weight": np.where(np.repeat(rng.random(R) < 0.3, 2), 0.0, 1.0), "cell_id": 0})
        shp = pd.DataFrame({"geoid": [f"r{i}" for i in range(R + 3)], "other": 1.0}, index=index)
        for zw in ("area", "nan"):
            w = SimpleNamespace(zero_weight=zw, weights=wdf, georegions=af.GeoRegions(shp))
            want = recipe(res, names, rid, labels, w)
            got = agg._assemble_frame(res, names, rid, labels, w)
            pd.testing.assert_frame_equal(got, want)
            merged = shp[["geoid"]].merge(want, left_index=True, right_on="region_id").drop(columns="region_id")
            pd.testing.assert_frame_equal(agg._merge_regions(got, w), merged)
            # the panel as a tensor (where it is when the kernels have run), with the region table named: a FULL panel carries the
            # table's id column straight away (attrs["_merged"]), anything else takes the two steps above — same frames either way
            import torch
            for panel in (res, np.where(np.isnan(res), 0.5, res)):
                want_t = recipe(panel, names, rid, labels, w)
                merged_t = shp[["geoid"]].merge(want_t, left_index=True, right_on="region_id").drop(columns="region_id")
                got_t = agg._assemble_frame(torch.from_numpy(panel), names, rid, labels, w, merge_with=w)
                direct = got_t.attrs.pop("_merged", False)
                assert direct == (not np.isnan(panel).any() and index.is_monotonic_increasing and 9999 not in rid), (trial, direct)
                pd.testing.assert_frame_equal(got_t if direct else agg._merge_regions(got_t, w), merged_t)


def test_cf_time_sel_is_month_and_day_granular():
    """`Dataset(time_sel=)` on a CF calendar selects like xarray's partial-date-string indexing (`dataset.py:90-91`): the
    whole month / day a string names, slices inclusive at both ends — and `io._time_window` (what the sharded store route
    trusts as the EXACT selection) returns the same run of steps, not whole years."""
    from aggfly_amd.cfcalendar import cf_range
    from aggfly_amd.io import _time_window
    t = cf_range("2000-01-01", 730, "D", "noleap")
    cube = np.arange(730 * 2 * 2, dtype=np.float64).reshape(730, 2, 2)
    mk = lambda sel: af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": t, "latitude": [0.0, 1.0], "longitude": [10.0, 11.0]}),
                                time_sel=sel)
    for sel, (k0, k1) in (("2000-06", (151, 181)), (slice("2000-03", "2000-09"), (59, 273)), ("2001", (365, 730)),
                          (slice("2000-12-31", "2001-01-02"), (364, 367)), (slice(None, "2000-01"), (0, 31))):
        ds = mk(sel)
        assert len(ds.time) == k1 - k0 and ds.time[0] == t[k0] and ds.time[len(ds.time) - 1] == t[k1 - 1], sel
        np.testing.assert_array_equal(ds.cube(), cube[k0:k1])
        assert _time_window(t, sel) == (k0, k1), sel
    assert _time_window(t, "1999") == (0, 0)
    with pytest.raises(ValueError, match="cannot parse"):
        t.sel_positions("June 2000")


def test_choose_backend_compares_with_the_local_world(monkeypatch):
    from aggfly_amd.distributed import choose_backend
    monkeypatch.setenv("WORLD_SIZE", "16")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert choose_backend(8) == "nccl"            # 2 nodes x 8 GPUs is an RCCL job
    assert choose_backend(1) == "gloo"
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert choose_backend(8) == "gloo" and choose_backend(16) == "nccl"
    assert choose_backend(1, "nccl") == "nccl"


def test_bench_refuses_ranks_without_their_own_gpu():
    """bench.py --gpus N: N ranks on fewer visible GPUs must not silently share a card and fall back to gloo — the run is
    refused (non-zero exit, one line, no JSON) unless AGGFLY_BENCH_BACKEND=gloo asks for a rehearsal, both as the plain
    command (which would start its own ranks) and as a rank under a launcher's environment."""
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("AGGFLY_BENCH_BACKEND", "AGGFLY_BENCH_DRY_LAUNCH", "WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "RCCL needs one GPU per rank" in r.stderr and "Traceback" not in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode != 0 and "RCCL needs one GPU per rank" in r.stderr and "AGGFLY_BENCH_BACKEND=gloo" in r.stderr
    assert r.stdout.strip() == ""                                     # no JSON line from a refused run
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE is 4" in r.stderr


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N ...` — the driver's command line, no torchrun around it — starts the N ranks itself as a CHILD
    `python -m torch.distributed.run` on 127.0.0.1 with a free port and the same arguments (AGGFLY_BENCH_DRY_LAUNCH shows the
    command and starts nothing); as a gloo rehearsal without any GPU the child's ranks really start and fail loudly at the
    engine's door (no CPU fallback), and the parent hands their exit code on."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("AGGFLY_BENCH_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    args = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    r = subprocess.run([sys.executable, bench] + args, capture_output=True, text=True, timeout=300, env=dict(env, AGGFLY_BENCH_DRY_LAUNCH="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout)["launch"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    at = cmd.index(bench)
    assert cmd[at + 1:] == args                                       # the ranks get the command line unchanged
    # one rank per GPU needs no launch at all
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--help"], capture_output=True, text=True, timeout=300, env=dict(env, AGGFLY_BENCH_DRY_LAUNCH="1"))
    assert r.returncode == 0 and "launch" not in r.stdout
    import torch
    if torch.cuda.is_available():
        return                                                        # (the GPU suite runs the real thing: tests/test_gpu_sharded.py)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--ny", "8", "--nx", "8", "--T", "48", "--regions", "3"],
                       capture_output=True, text=True, timeout=600, env=dict(env, AGGFLY_BENCH_BACKEND="gloo"))
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "no HIP device visible" in r.stderr or "No HIP GPUs" in r.stderr or "HipEngineError" in r.stderr, r.stderr[-1500:]


def test_plan_and_csr_caches_are_keyed_by_device(monkeypatch):
    """A handle lives on ONE device (include/aggfly_hip.h "Devices"): with a process (or a worker thread) per GPU the
    caches must never serve card 0's plan / weight table for a cube on card r.  The library calls are replaced by
    recorders here (no GPU in this suite); what is checked is the host logic: the device index is part of both cache keys
    and reaches the constructors."""
    made = []

    class FakePlan:
        def __init__(self, *a, device=None, **k):
            self.device_index = device
            made.append(("plan", device))

        def scratch_bytes(self):
            return 0

    class FakeCSR:
        def __init__(self, rows, cols, w, R, n_cells, device=None):
            self.device_index = device
            made.append(("csr", device))

    current = {"dev": 0}
    monkeypatch.setattr(hip, "FusedPlan", FakePlan)
    monkeypatch.setattr(hip, "CSR", FakeCSR)
    monkeypatch.setattr(hip, "_device_index", lambda d=None: current["dev"] if d is None else int(d))
    monkeypatch.setattr(eng, "_PLAN_CACHE", {})
    monkeypatch.setattr(eng, "_CSR_CACHE", {})
    ib, ob = np.arange(0, 49, 24), np.array([0, 2])
    cols = [dict(inner="mean", outer="sum")]
    p0 = eng.get_plan(48, 12, hip.F64, ib, ob, cols)                  # the calling thread's current device: 0
    assert eng.get_plan(48, 12, hip.F64, ib, ob, cols) is p0          # cached
    p3 = eng.get_plan(48, 12, hip.F64, ib, ob, cols, device=3)        # the same plan for a cube on card 3: a new handle there
    assert p3 is not p0 and p3.device_index == 3 and p0.device_index == 0
    current["dev"] = 5                                                # a thread whose current device is 5
    assert eng.get_plan(48, 12, hip.F64, ib, ob, cols).device_index == 5
    assert [k[0] for k in eng._PLAN_CACHE] == [0, 3, 5]

    arr = np.zeros((4, 3, 4))
    ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"],
                                 {"time": pd.date_range("2000-01-01", periods=4, freq="D"), "latitude": [1.0, 2.0, 3.0],
                                  "longitude": [10.0, 11.0, 12.0, 13.0]}))
    tab = pd.DataFrame({"cell_id": [0, 1, 5], "index_right": [0, 0, 1], "weight": [0.5, 0.5, 1.0]})
    w = af.weights_from_objects(ds, af.GeoRegions(pd.DataFrame({"geoid": ["a", "b"]})), table=tab)
    c5, _ = eng.get_csr(w, ds)
    c2, _ = eng.get_csr(w, ds, device=2)
    assert c5.device_index == 5 and c2.device_index == 2 and eng.get_csr(w, ds, device=2)[0] is c2
    assert eng.dataset_device(ds) == 5                                # a host array goes to the current device
    assert made.count(("csr", 2)) == 1 and made.count(("plan", 3)) == 1


def test_kernel_menu_shape():
    """The production menu (`make`) holds no tuning arm, every region-fused twin has its plain variant beside it (same fields, FEAT bit
    11 apart), twins exist only for direct-load two-level forms of at most six columns and four slots, the three-row (8-hourly) forms are
    lean short-group forms with depth 6; `MENU=arms` adds the arms and nothing else; the loaded library reports the menu it was built from."""
    sys.path.insert(0, os.path.join(ROOT, "aggfly_amd", "csrc"))
    import gen_variants as gv
    full, arms = gv.menu("full"), gv.menu("arms")
    assert all(v[8] for v in full) and sum(1 for v in arms if not v[8]) == len(arms) - len(full) > 0
    assert set(full) <= set(arms)
    keys = {v[:8] for v in full}
    twins = [v for v in full if v[7] & 2048]
    assert twins and all(v[:7] + (v[7] & ~2048,) in keys for v in twins)
    assert all(v[1] == 0 and v[5] <= 6 and v[4] <= 4 and not (v[7] & (8 | 16 | 32)) for v in twins)          # pipe, kmax, nthr, no tki / sl / hb
    tri = [v for v in full if v[7] & 4096]
    assert tri and all((v[7] & 128) and (v[7] & 256) and not (v[7] & 1024) and v[6] % 3 == 0 for v in tri)
    assert len({gv.name_of(v) for v in arms}) == len(arms)                                                   # names are unique
    info = hip.build_info()
    assert info["abi"] == hip.ABI_VERSION and info["menu"] in ("full", "arms", "dev")
    if info["menu"] == "full":
        assert info["variants"] == len(full) and info["arms"] == 0 and info["region_fused_twins"] == len(twins)


def test_every_name_the_reference_package_exports_exists_here():
    """`aggfly/__init__.py:1-27` re-exports 27 names; a script written against it must import unchanged (`import aggfly_amd as af`).
    The weights PRODUCERS (geopandas / rasterio work, CPU-side per the north_star) exist as names that point to the supported way in:
    the cached table through `weights_from_objects(table=)` / `weights_from_feather`."""
    names = ["TemporalAggregator", "SpatialAggregator", "aggregate_dataset", "aggregate_time", "aggregate_space", "distributed_client",
             "is_distributed", "start_dask_client", "shutdown_dask_client", "Dataset", "Grid", "dataset_from_path", "dataset_to_zarr",
             "zarr_from_path", "CropWeights", "PopWeights", "GridWeights", "SecondaryWeights", "weights_from_objects", "pop_weights_from_path",
             "crop_weights_from_path", "secondary_weights_from_path", "GeoRegions", "georegions_from_path", "georegions_from_gdf", "shapefile_info"]
    missing = [n for n in names if not hasattr(af, n)]
    assert not missing, missing
    for n in ("CropWeights", "PopWeights", "SecondaryWeights"):
        with pytest.raises(NotImplementedError, match="weights_from_objects"):
            getattr(af, n)("raster.tif")
    for n in ("pop_weights_from_path", "crop_weights_from_path", "secondary_weights_from_path", "shapefile_info"):
        with pytest.raises(NotImplementedError, match="weights_from_objects"):
            getattr(af, n)("x")


def test_gpu_decode_batch_plan(monkeypatch):
    """`io._decode_batches`: how a decode-in-HBM request is cut (profiles/r04_ingest_batches.txt): equal batches of ~144 MB decoded, at most 16
    of them and 512 MB each, every chunk in exactly one batch or in the host-decoded tail; the tail (<= 128 MB, <= a fifth of the request) only
    for requests of 256 MB and more on chunks of whole time steps; the knobs override."""
    from aggfly_amd import io as afio
    for k in ("AGGFLY_HIP_GPU_DECODE_BATCH_MB", "AGGFLY_HIP_GPU_DECODE_CUTS", "AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MB", "AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MIN_MB",
              "AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MAX_PCT"):
        monkeypatch.delenv(k, raising=False)
    cb, nblk = 24 * 104 * 236 * 4, 37                                  # the BASELINE configs[0] store's chunks: 2.36 MB, 37 Blosc blocks
    for n in (1, 3, 6, 100, 146, 365, 1460, 5000):
        for whole in (True, False):
            cuts, per, n_tail = afio._decode_batches(n, cb, nblk, whole)
            assert cuts[0] == 0 and cuts[-1] == n - n_tail and cuts == sorted(set(cuts)), (n, cuts)
            sizes = np.diff(cuts)
            assert per == sizes.max() and sizes.max() - sizes.min() <= 1, (n, sizes)             # equal batches
            assert len(sizes) <= 16 or per * cb >= 0.95 * (512 << 20), (n, len(sizes))           # at most 16, unless the 512 MB cap binds
            assert per * cb <= (512 << 20) + cb
            assert n_tail == (0 if (not whole or n * cb < (256 << 20)) else min((128 << 20) // cb, n // 5)), (n, whole, n_tail)
    assert len(afio._decode_batches(365, cb, nblk, True)[0]) - 1 == 5 and afio._decode_batches(365, cb, nblk, True)[2] == 56    # 0.86 GB: 5 x 62 chunks + 56
    assert len(afio._decode_batches(365, cb, nblk, False)[0]) - 1 == 6                                                          # space-tiled: 6 batches, no tail
    big = 265 << 20                                                    # the converter's whole-series tiles: one chunk per batch, no tail
    assert afio._decode_batches(6, big, 4200, False) == ([0, 1, 2, 3, 4, 5, 6], 1, 0)
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MIN_MB", "0")   # (tests / the ingest fuzzer: a tail on small requests and tiled chunks too)
    assert afio._decode_batches(10, cb, nblk, False)[2] == 2 and afio._decode_batches(10, cb, nblk, True)[2] == 2
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MB", "0")
    assert afio._decode_batches(365, cb, nblk, True)[2] == 0
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_CUTS", "0.25,0.5")
    assert afio._decode_batches(100, cb, nblk, True) == ([0, 25, 50, 100], 50, 0)
    monkeypatch.delenv("AGGFLY_HIP_GPU_DECODE_CUTS")
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_BATCH_MB", "1")
    cuts, per, _ = afio._decode_batches(10, cb, nblk, True)
    assert per == 1 and cuts == list(range(11))
