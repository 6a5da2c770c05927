"""GPU parity through the public API (``import aggfly_amd as af``), written to read like the
reference's own tests (`aggfly/tests/test_aggregate.py`): same fixtures, same specs, same
assertions — checked against the committed golden vectors and against the oracle."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_inputs as gi  # noqa: E402

import aggfly_amd as af
from aggfly_amd import synth
from oracle import ref_aggregate as ra
from oracle.ref_calendar import cf_daily_index
from oracle.ref_spatial import wavg_loops

pytestmark = pytest.mark.gpu
G = gi.goldens()


def _xr(arr, time, lat, lon):
    return af.DataArray(data=arr, dims=["time", "latitude", "longitude"],
                        coords={"time": time, "latitude": lat, "longitude": lon})


@pytest.fixture(name="dataset_360")
def dataset_360_fixture():
    arr, time, lat, lon = gi.dataset_360_inputs()
    return af.Dataset(_xr(arr, time, lat, lon), lon_is_360=True)


@pytest.fixture(name="georegion")
def georegion_fixture():
    return af.GeoRegions(pd.DataFrame({"geoid": ["region_1"]}), regionid="geoid")


@pytest.fixture(name="weights")
def weights_fixture(dataset_360, georegion):
    # the table test_weights pins (test_aggregate.py:234-237), sorted by cell_id (:176)
    w = af.weights_from_objects(dataset_360, georegion, table=gi.g2_weights_table())
    w.calculate_weights()
    w.weights = w.weights.sort_values("cell_id")
    return w


def _table(adict):
    return np.stack([adict[k].da.transpose("latitude", "longitude", "time").values.reshape(-1) for k in adict], axis=1)


@pytest.mark.parametrize("engine", ["auto", "hip", "numba"])
def test_aggregate_time(torch_cuda, dataset_360, weights, engine):
    adict = af.aggregate_time(dataset=dataset_360, weights=weights, engine=engine, **gi.g1_spec())
    assert list(adict) == G["G1_temporal_table"]["columns"]
    assert np.allclose(_table(adict), np.array(G["G1_temporal_table"]["values"]))


def test_aggregate(torch_cuda, dataset_360, weights):
    df = af.aggregate_dataset(dataset=dataset_360, weights=weights, **gi.g2_spec())
    assert list(df.columns) == ["geoid", "time", "tavg_1", "tavg_2"]
    assert np.allclose(df[["tavg_1", "tavg_2"]].values, np.array(G["G2_panel"]["values"]))
    assert df["time"].iloc[0] == pd.Timestamp("2000-07-31") and df["geoid"].iloc[0] == "region_1"


def test_aggregate_dataset_deprecated_cluster_kwargs(torch_cuda, dataset_360, weights):
    spec = dict(tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "month"})])
    ref = af.aggregate_dataset(dataset=dataset_360.deepcopy(), weights=weights, **spec)
    with pytest.warns(DeprecationWarning, match="no longer builds a Dask cluster"):
        got = af.aggregate_dataset(dataset=dataset_360.deepcopy(), weights=weights,
                                   n_workers=50, processes=True, cluster_args={}, **spec)
    assert "tavg" in got.columns and "n_workers" not in got.columns
    assert np.allclose(got["tavg"].values, ref["tavg"].values, equal_nan=True)
    with pytest.raises(ValueError, match="No dataset provided"):
        af.aggregate_dataset(weights=weights, **spec)
    with pytest.raises(ValueError, match="engine must be"):
        af.aggregate_dataset(dataset=dataset_360, weights=weights, engine="bogus", **spec)


def test_sine_dd_partial_nan_masking(torch_cuda):
    time = pd.date_range("2000-07-01", periods=4, freq="12h")
    lat, lon = np.array([-45.0, 45.0]), np.array([10.0, 100.0])
    arr = np.empty((4, 2, 2), dtype="float64")
    arr[0], arr[1], arr[2], arr[3] = 15.0, 30.0, 18.0, 28.0
    arr[1, 0, 1] = np.nan
    arr[0, 1, 0] = np.nan
    out = af.aggregate_time(dataset=af.Dataset(_xr(arr.copy(), time, lat, lon), lon_is_360=False), weights=None,
                            cdd=[("aggregate", {"calc": "sine_dd", "groupby": "date", "ddargs": [20, 99, 0]})])
    got = out["cdd"].da.transpose("latitude", "longitude", "time").values
    want = ra.aggregate_time(ra.ODataset(arr, time, lat, lon, False),
                             {"cdd": [("aggregate", {"calc": "sine_dd", "groupby": "date", "ddargs": [20, 99, 0]})]})["cdd"].values
    assert np.allclose(got, np.transpose(want, (1, 2, 0)), rtol=1e-10, atol=1e-12, equal_nan=True)
    assert np.isnan(got[0, 1, 0]) and np.isnan(got[1, 0, 0])
    assert np.isfinite(got[0, 0, 0]) and got[0, 0, 0] > 0
    assert np.isfinite(got[0, 1, 1]) and got[0, 1, 1] > 0


def _cf_dataset(calendar, ndays, nan=False, seed=0):
    arr, lat, lon = gi.cftime_cube(ndays, nan=nan, seed=seed)
    return (af.Dataset(_xr(arr, af.cf_range("2000-01-01", ndays, "D", calendar), lat, lon), lon_is_360=False),
            ra.ODataset(arr, cf_daily_index(calendar, ndays), lat, lon, False))


@pytest.mark.parametrize("calendar", ["360_day", "noleap"])
@pytest.mark.parametrize("nan", [False, True])
def test_cftime_parity_with_oracle(torch_cuda, calendar, nan):
    ds, ods = _cf_dataset(calendar, 720, nan=nan)
    for name, steps in gi.k3_specs().items():
        got = af.aggregate_time(dataset=ds.deepcopy(), weights=None, v=steps)
        want = ra.aggregate_time(ods, {"v": steps}, engine="numba")
        assert set(got) == set(want)
        for k in got:
            a = got[k].da.transpose("time", "latitude", "longitude").values
            assert np.allclose(a, want[k].values, rtol=1e-10, atol=1e-10, equal_nan=True), (calendar, nan, name, k)
            if "sine" not in name:
                np.testing.assert_array_equal(a, want[k].values)
            assert len(got[k].time) == len(want[k].time)


def test_cftime_empty_bin(torch_cuda):
    t = af.cf_range("2000-01-01", 90, "D", "360_day")
    keep = np.nonzero(t.fields()[1] != 2)[0]
    arr = np.random.default_rng(1).normal(15, 10, (len(keep), 2, 2))
    ds = af.Dataset(_xr(arr, t[keep], [-45.0, 45.0], [10.0, 100.0]), lon_is_360=False)
    a = af.aggregate_time(dataset=ds, weights=None, v=[("aggregate", {"calc": "mean", "groupby": "month"})])["v"]
    a = a.da.transpose("latitude", "longitude", "time").values
    assert a.shape[-1] == 3 and np.all(np.isnan(a[..., 1])) and np.all(np.isfinite(a[..., [0, 2]]))


def test_cftime_end_to_end_aggregate_dataset(torch_cuda, weights):
    lon = np.array([90.0, 270.0]); lat = np.array([-45.0, 45.0])
    arr = np.random.default_rng(3).normal(20, 15, (360, 2, 2))
    spec = dict(tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "month"})])
    got = af.aggregate_dataset(dataset=af.Dataset(_xr(arr.copy(), af.cf_range("2000-01-01", 360, "D", "360_day"), lat, lon), lon_is_360=True),
                               weights=weights, **spec)
    ow = ra.OWeights(gi.g2_weights_table(), np.arange(4), pd.Series(["region_1"], index=[0]), "geoid", "nan")
    want = ra.aggregate_dataset(ow, ra.ODataset(arr, cf_daily_index("360_day", 360), lat, lon, True), engine="dask", **spec)
    assert len(got) == 12
    assert getattr(got["time"].iloc[0], "calendar", None) == "360_day"
    assert np.allclose(got["tavg"].values, want["tavg"].values, rtol=1e-12, equal_nan=True)


def test_cftime_week_groupby_raises(torch_cuda):
    da = _xr(np.random.rand(60, 2, 2), af.cf_range("2000-01-01", 60, "D", "360_day"), [-45.0, 45.0], [10.0, 100.0])
    with pytest.raises(NotImplementedError, match="week"):
        af.aggregate_time(dataset=af.Dataset(da, lon_is_360=False), weights=None,
                          v=[("aggregate", {"calc": "mean", "groupby": "week"})])
    with pytest.raises(NotImplementedError, match="week"):
        af.TemporalAggregator("mean", "week").execute(af.Dataset(da, lon_is_360=False))


# ---- spatial stage through the real SpatialAggregator with minimal stand-ins (test_aggregate.py:565-664)
def _run_spatial(vals, time, lat, lon, wdf, names):
    grid = SimpleNamespace(cell_id=np.array([0, 1, 2, 3]))
    weights = SimpleNamespace(grid=grid, weights=wdf)
    dlist = [af.Dataset(_xr(vals, time, lat, lon), lon_is_360=False) for _ in names]
    return af.SpatialAggregator(dlist, weights, names=names).compute()


@pytest.mark.parametrize("case", ["multiregion_nan", "dropna_empty_group"])
def test_spatial_matmul_vs_loop_oracle(torch_cuda, case):
    vals, time, lat, lon, wdf = gi.k7_case(case)
    names = ["v"]
    oracle = wavg_loops({"v": vals.reshape(len(time), 4).T}, time.values, [0, 1, 2, 3], wdf, names)
    got = _run_spatial(vals, time, lat, lon, wdf, names).sort_values(["region_id", "time"]).reset_index(drop=True)
    oracle = oracle.sort_values(["region_id", "time"]).reset_index(drop=True)
    assert got.shape == oracle.shape
    assert (got[["region_id", "time"]].values == oracle[["region_id", "time"]].values).all()
    assert np.allclose(got["v"].values, oracle["v"].values, equal_nan=True)


def _two_regions_one_empty(arr=None, zero_weight="nan"):
    """test_aggregate.py:1430-1450 with the weights table the reference would produce."""
    lat = np.arange(0, 4.0) + 0.5
    lon = np.arange(0, 4.0) + 0.5
    arr = np.ones((2, 4, 4)) if arr is None else arr
    ds = af.Dataset(_xr(arr, pd.date_range("2000-01-01", periods=2), lat, lon), lon_is_360=False)
    gr = af.GeoRegions(pd.DataFrame({"geoid": ["has_pop", "no_pop"]}), regionid="geoid")
    cells = np.arange(16)
    right = (cells % 4) >= 2
    if zero_weight == "drop":
        keep = ~right
        tab = pd.DataFrame({"cell_id": cells[keep], "index_right": 0, "weight": 1 / 8})
    else:
        w = np.where(right, 0.0 if zero_weight == "nan" else 1 / 8, 1 / 8)
        tab = pd.DataFrame({"cell_id": cells, "index_right": right.astype(int), "weight": w})
    return ds, af.weights_from_objects(ds, gr, table=tab, zero_weight=zero_weight)


def _panel(ds, w):
    return af.aggregate_dataset(dataset=ds, weights=w, tavg=[("aggregate", {"calc": "mean", "groupby": "date"})])


def test_zero_weight_policies_reach_the_panel(torch_cuda):
    ds, w = _two_regions_one_empty()
    df = _panel(ds, w)
    assert set(df.geoid) == {"has_pop", "no_pop"}
    assert df.loc[df.geoid == "no_pop", "tavg"].isna().all() and df.loc[df.geoid == "has_pop", "tavg"].notna().all()
    ds, w = _two_regions_one_empty(zero_weight="area")
    df = _panel(ds, w)
    assert set(df.geoid) == {"has_pop", "no_pop"} and df.tavg.notna().all()
    ds, w = _two_regions_one_empty(zero_weight="drop")
    assert set(_panel(ds, w).geoid) == {"has_pop"}
    arr = np.ones((2, 4, 4)); arr[1] = np.nan
    ds, w = _two_regions_one_empty(arr)
    df = _panel(ds, w)
    assert len(df[df.geoid == "has_pop"]) == 1
    empty = df[df.geoid == "no_pop"]
    assert len(empty) == 2 and empty.tavg.isna().all()


# ---- mid-size seeded parity of the whole path against the oracle
def _mid_case(dtype, T=24 * 75 + 3, ny=12, nx=20, R=13, lon360=False):
    cube = synth.temperature_cube(T, ny, nx, dtype=dtype, seed=21, ocean_frac=0.08, scattered_nan=40)
    time = pd.date_range("2001-01-20", periods=T, freq="h")
    lat = 30.0 + 0.25 * np.arange(ny)
    lon = (200.0 if lon360 else -120.0) + 0.25 * np.arange(nx)
    tab = synth.weights_table(ny, nx, R, seed=22, secondary=True, zero_frac=0.1)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}), regionid="geoid")
    ds = af.Dataset(_xr(cube, time, lat, lon), lon_is_360=lon360)
    w = af.weights_from_objects(ds, gr, table=tab)
    ods = ra.ODataset(cube.astype(np.float64), time, lat, lon, lon360)
    ow = ra.OWeights(tab, np.arange(ny * nx), gr.shp["geoid"], "geoid", "nan")
    return ds, w, ods, ow


C2_SPEC = dict(
    dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "month"})],
    tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 5)}),
          ("aggregate", {"calc": "sum", "groupby": "month"})],
)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("lon360", [False, True])
def test_whole_path_c2_against_oracle(torch_cuda, dtype, lon360):
    ds, w, ods, ow = _mid_case(dtype, lon360=lon360)
    got = af.aggregate_dataset(dataset=ds, weights=w, **C2_SPEC)
    want = ra.aggregate_dataset(ow, ods, engine="numba", **C2_SPEC)
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    assert (got["geoid"].values == want["geoid"].values).all() and (got["time"].values == want["time"].values).all()
    cols = [c for c in got.columns if c not in ("geoid", "time")]
    # north_star: fp64 results within 1e-10 relative of the reference CPU path (f32 storage
    # is compared with the reference run on the float64-cast input)
    np.testing.assert_allclose(got[cols].values, want[cols].values, rtol=1e-10, atol=0, equal_nan=True)


def test_staged_path_three_levels_and_raw_transform(torch_cuda):
    ds, w, ods, ow = _mid_case(np.float64)
    spec = dict(
        a=[("aggregate", {"calc": "max", "groupby": "date"}), ("aggregate", {"calc": "mean", "groupby": "month"}),
           ("aggregate", {"calc": "sum", "groupby": "year"})],
        b=[("transform", {"transform": "power", "exp": np.arange(2, 3)}), ("aggregate", {"calc": "mean", "groupby": "date"}),
           ("aggregate", {"calc": "sum", "groupby": "year"})],
        c=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "nanmean", "groupby": "year"})],
    )
    got = af.aggregate_dataset(dataset=ds, weights=w, **spec)
    want = ra.aggregate_dataset(ow, ods, engine="numba", **spec)
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    cols = ["a", "b_2", "c"]
    np.testing.assert_allclose(got[cols].values, want[cols].values, rtol=1e-10, equal_nan=True)


def test_mixed_frequencies_share_validity(torch_cuda):
    """Two names with different inner groupings run as two passes; validity stays shared."""
    ds, w, ods, ow = _mid_case(np.float64)
    spec = dict(
        m=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "month"})],
        x=[("aggregate", {"calc": "max", "groupby": "month"})],
        s=[("aggregate", {"calc": "sine_dd", "groupby": "date", "ddargs": [[10, 30, 0], [0, 12, 1]]}),
           ("aggregate", {"calc": "sum", "groupby": "month"})],
    )
    got = af.aggregate_dataset(dataset=ds, weights=w, **spec)
    want = ra.aggregate_dataset(ow, ods, engine="numba", **spec)
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    cols = [c for c in got.columns if c not in ("geoid", "time")]
    np.testing.assert_allclose(got[cols].values, want[cols].values, rtol=1e-10, atol=1e-12, equal_nan=True)


def test_temporal_aggregator_class_and_device_resident_dataset(torch_cuda):
    ds, w, ods, ow = _mid_case(np.float32)
    ds.to_device()
    out = af.TemporalAggregator("bins", "month", ddargs=[[0, 15, 0], [15, 30, 0]]).execute(ds)
    assert isinstance(out, list) and len(out) == 2
    want = ra.OTemporalAggregator("bins", "month", ddargs=[[0, 15, 0], [15, 30, 0]]).execute(ods)
    for o, wv in zip(out, want):
        np.testing.assert_array_equal(o.da.transpose("time", "latitude", "longitude").values, wv.values)
    one = af.TemporalAggregator("mean", "date").execute(ds)
    np.testing.assert_array_equal(one.da.transpose("time", "latitude", "longitude").values,
                                  ra.OTemporalAggregator("mean", "date").execute(ods).values)


def test_daily_data_two_level_specs_collapse_to_one_level(torch_cuda):
    """CMIP6-style daily data (configs[3]): 'mean@date' is the identity there, so the engine
    runs bins/dd/sum@year straight on the raw steps.  Same numbers as the two-level oracle."""
    T, ny, nx = 365 * 3, 10, 16
    cube = synth.temperature_cube(T, ny, nx, seed=41, steps_per_day=1, ocean_frac=0.1, scattered_nan=30)
    lat, lon = -20 + 2.0 * np.arange(ny), 2.5 * np.arange(nx)
    edges = np.arange(-20, 50, 5.0)
    spec = dict(
        tbin=[("aggregate", {"calc": "mean", "groupby": "date"}),
              ("aggregate", {"calc": "bins", "groupby": "year", "ddargs": [[a, b, 0] for a, b in zip(edges[:-1], edges[1:])]})],
        gdd=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "dd", "groupby": "year", "ddargs": [10, 30, 0]})],
        tsum=[("aggregate", {"calc": "max", "groupby": "date"}), ("aggregate", {"calc": "mean", "groupby": "year"})],
        t2=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(2, 3)}),
            ("aggregate", {"calc": "sum", "groupby": "year"})],
    )
    tab = synth.weights_table(ny, nx, 7, seed=42)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
    for time, otime in ((pd.date_range("2001-01-01", periods=T, freq="D"),) * 2,
                        (af.cf_range("2001-01-01", T, "D", "noleap"), cf_daily_index("noleap", T, (2001, 1, 1)))):
        ds = af.Dataset(_xr(cube, time, lat, lon), lon_is_360=True)
        w = af.weights_from_objects(ds, gr, table=tab)
        got = af.aggregate_dataset(dataset=ds, weights=w, **spec)
        ow = ra.OWeights(tab, np.arange(ny * nx), gr.shp["geoid"], "geoid", "nan")
        want = ra.aggregate_dataset(ow, ra.ODataset(cube, otime, lat, lon, True), engine="numba", **spec)
        assert list(got.columns) == list(want.columns) and len(got) == len(want) == 3 * len(gr.shp)
        cols = [c for c in got.columns if c not in ("geoid", "time")]
        np.testing.assert_allclose(got[cols].values, want[cols].values, rtol=1e-12, atol=0, equal_nan=True)


def test_interact_and_spline_transforms(torch_cuda):
    """X2: `inter` (element-wise product with a second dataset, dataset.py:483-563) and `spline` (hinge at 20,
    dataset.py:475-481) are both fused into the streaming kernel's group end: ONE pass.  Against the oracle."""
    from aggfly_amd import engine as eng
    ds, w, ods, ow = _mid_case(np.float64)
    assert eng.lower_spec("tx", [("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "inter", "inter": ds}),
                                 ("aggregate", {"calc": "sum", "groupby": "month"})])[1] is True
    passes = []
    real = eng._run_fused_pass
    eng._run_fused_pass = lambda *a, **k: passes.append(1) or real(*a, **k)
    daily = af.aggregate_time(dataset=ds, weights=None, p=[("aggregate", {"calc": "max", "groupby": "date"})])["p"]
    odaily = ra.aggregate_time(ods, {"p": [("aggregate", {"calc": "max", "groupby": "date"})]})["p"]
    spec = lambda other: dict(
        tx=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "inter", "inter": other}),
            ("aggregate", {"calc": "sum", "groupby": "month"})],
        sp=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "spline"}),
            ("aggregate", {"calc": "sum", "groupby": "month"})])
    try:
        n0 = len(passes)
        got = af.aggregate_dataset(dataset=ds, weights=w, **spec(daily))
        assert len(passes) - n0 == 1                                        # inter + spline + their sums: one fused pass
    finally:
        eng._run_fused_pass = real
    want = ra.aggregate_dataset(ow, ods, engine="numba", **spec(odaily))
    assert list(got.columns) == list(want.columns) == ["geoid", "time", "tx", "sp_spline1", "sp_spline2"]
    assert len(got) == len(want)
    np.testing.assert_allclose(got[["tx", "sp_spline1", "sp_spline2"]].values, want[["tx", "sp_spline1", "sp_spline2"]].values,
                               rtol=1e-12, equal_nan=True)
    # the second array as a bare numpy array in the step output's (time, lat, lon) layout (the reference's compiled engine
    # leaves its outputs so, nb_kernels.py:293), and a wrong shape -> the reference's AssertionError
    got2 = af.aggregate_dataset(dataset=ds, weights=w, **spec(odaily.values))
    np.testing.assert_array_equal(got2["tx"].values, got["tx"].values)
    with pytest.raises(AssertionError):
        af.aggregate_dataset(dataset=ds, weights=w, **spec(np.ascontiguousarray(np.moveaxis(odaily.values, 0, -1))))
    # inter on RAW data (before any aggregate) and after the outer level stay staged — through hip.transform
    spec3 = dict(r=[("transform", {"transform": "inter", "inter": ds}), ("aggregate", {"calc": "mean", "groupby": "month"})])
    ospec3 = dict(r=[("transform", {"transform": "inter", "inter": ods}), ("aggregate", {"calc": "mean", "groupby": "month"})])
    g3, w3 = af.aggregate_dataset(dataset=ds, weights=w, **spec3), ra.aggregate_dataset(ow, ods, engine="numba", **ospec3)
    np.testing.assert_allclose(g3["r"].values, w3["r"].values, rtol=1e-12, equal_nan=True)


def test_dataset_transforms_run_in_the_hip_library(torch_cuda):
    """`Dataset.power / spline / interact` (dataset.py:442-518) run `afhip_transform` — host arrays are uploaded, nothing is
    computed in numpy or torch — and follow numpy's dtype rules: float32 ** python int stays float32, ** np.int64 -> float64."""
    from aggfly_amd import hip
    rng = np.random.default_rng(8)
    arr = rng.normal(18, 9, (30, 5, 7))
    arr[3, 1, 2] = np.nan
    time = pd.date_range("2001-01-01", periods=30, freq="D")
    mk = lambda a: af.Dataset(af.DataArray(a, ["time", "latitude", "longitude"], {"time": time, "latitude": np.arange(5.0), "longitude": np.arange(7.0)}))
    calls = []
    real = hip.transform
    hip.transform = lambda *a, **k: calls.append(a[1]) or real(*a, **k)
    try:
        for dt in (np.float64, np.float32):
            a = arr.astype(dt)
            ds = mk(a)                                                     # host-resident: uploaded by the transform
            for e in (np.int64(2), np.int64(3), 2, -1, 0.5, np.float64(1.5)):
                got = ds.power(e)
                want = np.power(a, e)
                assert str(got.da.dtype).endswith(str(want.dtype)), (dt, e, got.da.dtype, want.dtype)
                # numpy's own pow is within an ulp, not correctly rounded (SVML on AVX-512 hosts): bit-equal only for x * x
                tol = 0 if want.dtype == np.float64 and float(e) == 2 else (3e-7 if want.dtype == np.float32 else 4e-16)
                np.testing.assert_allclose(got.cube().cpu().numpy(), want, rtol=tol, atol=0, equal_nan=True)
            s1, s2 = ds.spline()
            assert s1 is ds
            np.testing.assert_array_equal(s2.cube().cpu().numpy(), (a > 20) * (a - 20))
            other = rng.normal(1, 0.5, (5, 7, 30)).astype(dt)            # (lat, lon, time): the dataset's own layout
            got = ds.interact(other)
            np.testing.assert_array_equal(got.cube().cpu().numpy(), a * np.moveaxis(other, -1, 0))
            assert str(got.da.dtype).endswith(str(a.dtype))
            got = ds.interact(mk(np.moveaxis(other, -1, 0).astype(np.float64)))         # a Dataset, float64: promotes
            np.testing.assert_array_equal(got.cube().cpu().numpy(), a.astype(np.float64) * np.moveaxis(other, -1, 0).astype(np.float64))
            with pytest.raises(AssertionError):
                ds.interact(other[:, :, :5])
            d2 = mk(a)
            assert d2.power(2, update=True) is None and d2.history[-1] == "power2"
            np.testing.assert_allclose(d2.cube().cpu().numpy(), np.power(a, 2), rtol=3e-7 if dt == np.float32 else 0, equal_nan=True)
    finally:
        hip.transform = real
    assert calls.count("pow") == 14 and calls.count("hinge") == 2 and calls.count("inter") == 4


def test_non_integer_exponent_is_fused(torch_cuda):
    """`np.power` with a non-integer exponent (dataset.py:543) goes through the kernel's pow(): mean@date -> x ** 1.5 / 0.5
    -> sum@month in one pass, against the oracle."""
    ds, w, ods, ow = _mid_case(np.float64)
    spec = dict(t=[("aggregate", {"calc": "max", "groupby": "date"}), ("transform", {"transform": "power", "exp": [np.array([0.5, 1.5, 2.0])]}),
                   ("aggregate", {"calc": "sum", "groupby": "month"})])
    got = af.aggregate_dataset(dataset=ds, weights=w, **spec)
    want = ra.aggregate_dataset(ow, ods, engine="numba", **spec)
    cols = ["t_0.5", "t_1.5", "t_2.0"]
    assert list(got.columns)[2:] == cols == list(want.columns)[2:] and len(got) == len(want)
    np.testing.assert_allclose(got[cols].values, want[cols].values, rtol=1e-12, equal_nan=True)


def test_zarr_streams_straight_into_hbm(torch_cuda, tmp_path):
    """N2: a float Zarr store decoded chunk-parallel into pinned slabs and uploaded while the
    next slab decodes gives the same cube (and the same panel) as the host path."""
    T, ny, nx = 24 * 40, 9, 14
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=51, scattered_nan=20) + np.float32(273.15)
    time = pd.date_range("2002-01-01", periods=T, freq="h")
    lat, lon = 30 + 0.5 * np.arange(ny), 200 + 0.5 * np.arange(nx)
    ds = af.Dataset(_xr(cube, time, lat, lon), lon_is_360=True)
    store = str(tmp_path / "t.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 100, "latitude": 4, "longitude": 5})
    host = af.dataset_from_path(store, "t2m", preprocess=lambda x: x - 273.15)
    dev = af.dataset_from_path(store, "t2m", preprocess=lambda x: x - 273.15, device="cuda")
    assert dev.cube().is_cuda and dev.time.equals(time)
    np.testing.assert_array_equal(dev.cube().cpu().numpy(), host.cube())
    tab = synth.weights_table(ny, nx, 5, seed=52)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
    spec = dict(t=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "month"})])
    a = af.aggregate_dataset(dataset=dev, weights=af.weights_from_objects(dev, gr, table=tab), **spec)
    b = af.aggregate_dataset(dataset=host, weights=af.weights_from_objects(host, gr, table=tab), **spec)
    pd.testing.assert_frame_equal(a, b)
    # time-contiguous layout (each chunk = a run of whole time steps): the slab's Blosc chunks are read and
    # decoded by the native codec on an OpenMP team straight into the staging rows; 960 = 13 x 70 + 50, so
    # the last chunk is padded; zlib and raw stores take the generic per-chunk route
    for comp, fmt in (("blosc", 2), ("zlib", 2), (False, 2), ("blosc", 3), ("zstd", 3)):
        store2 = str(tmp_path / f"rows_{comp}_{fmt}.zarr")
        af.dataset_to_zarr(ds, store2, var="t2m", chunks={"time": 70, "latitude": ny, "longitude": nx}, compress=comp, zarr_format=fmt)
        dev2 = af.dataset_from_path(store2, "t2m", preprocess=lambda x: x - 273.15, device="cuda")
        np.testing.assert_array_equal(dev2.cube().cpu().numpy(), host.cube())
    # format-3 shards: the inner chunks are located through each shard's index and decoded by byte range
    for comp, chunks, shards in (("zstd", {"time": 30, "latitude": ny, "longitude": nx}, {"time": 240, "latitude": ny, "longitude": nx}),
                                 ("blosc", {"time": 100, "latitude": 4, "longitude": 5}, {"time": 300, "latitude": 8, "longitude": 10})):
        store3 = str(tmp_path / f"shard_{comp}.zarr")
        af.dataset_to_zarr(ds, store3, var="t2m", chunks=chunks, shards=shards, compress=comp, zarr_format=3)
        dev3 = af.dataset_from_path(store3, "t2m", preprocess=lambda x: x - 273.15, device="cuda")
        np.testing.assert_array_equal(dev3.cube().cpu().numpy(), host.cube())


def test_time_selection_reads_only_its_chunks(torch_cuda, tmp_path, monkeypatch):
    """time_sel on the streaming route decodes just the chunks that hold the selected steps (window not
    aligned to chunk edges, both chunk layouts, datetime and noleap calendars) and equals the host route."""
    from aggfly_amd import codec, io as afio
    T, ny, nx = 24 * 90, 6, 10
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=61, scattered_nan=15)
    lat, lon = 30 + 0.5 * np.arange(ny), 200 + 0.5 * np.arange(nx)
    calls = []
    real = codec.decode_ranges
    monkeypatch.setattr(codec, "decode_ranges", lambda kind, locs, outs, threads=8: calls.append(len(locs)) or real(kind, locs, outs, threads))
    for label, time, sel, nsel in (("dt", pd.date_range("2002-01-01", periods=T, freq="h"), slice("2002-02-03", "2002-02-20"), 18 * 24),
                                   ("cf", af.cf_range("1999-06-01", T, "D", "noleap"), slice("2001", "2002"), 730)):
        ds = af.Dataset(_xr(cube, time, lat, lon), lon_is_360=True)
        for chunks in ({"time": 100, "latitude": ny, "longitude": nx}, {"time": 500, "latitude": 4, "longitude": 5}):
            store = str(tmp_path / f"w_{label}_{chunks['time']}.zarr")
            af.dataset_to_zarr(ds, store, var="t2m", chunks=chunks)
            host = af.dataset_from_path(store, "t2m", time_sel=sel)
            calls.clear()
            dev = af.dataset_from_path(store, "t2m", time_sel=sel, device="cuda")
            assert dev.cube().is_cuda and len(dev.time) == len(host.time) == nsel
            assert list(dev.time) == list(host.time)
            np.testing.assert_array_equal(dev.cube().cpu().numpy(), host.cube())
            n_all = -(-T // chunks["time"]) * -(-ny // chunks["latitude"]) * -(-nx // chunks["longitude"])
            assert 0 < sum(calls) < n_all, (sum(calls), n_all)                       # fewer chunk files than the store holds


def test_multi_file_dataset_streams_every_store(torch_cuda, tmp_path):
    """A list / glob of per-year stores goes through the streaming route store by store and is joined along
    time in HBM, like the host route's open_mfdataset-style concatenation (preprocess applied either way)."""
    ny, nx = 7, 9
    lat, lon = 30 + 0.5 * np.arange(ny), 200 + 0.5 * np.arange(nx)
    cubes = []
    for i, year in enumerate((2001, 2002, 2003)):
        T = 24 * (10 + i)
        cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=80 + i, scattered_nan=5) + np.float32(273.15)
        cubes.append(cube)
        ds = af.Dataset(_xr(cube, pd.date_range(f"{year}-01-01", periods=T, freq="h"), lat, lon), lon_is_360=True)
        af.dataset_to_zarr(ds, str(tmp_path / f"era_{year}.zarr"), var="t2m", chunks={"time": 50, "latitude": ny, "longitude": nx})
    pattern = str(tmp_path / "era_*.zarr")
    pre = lambda x: x - 273.15
    host = af.dataset_from_path(pattern, "t2m", preprocess=pre)
    dev = af.dataset_from_path(pattern, "t2m", preprocess=pre, device="cuda")
    assert dev.cube().is_cuda and dev.cube().shape == (24 * 33, ny, nx) and dev.time.equals(host.time)
    np.testing.assert_array_equal(dev.cube().cpu().numpy(), host.cube())
    dev2 = af.dataset_from_path(pattern, "t2m", preprocess=pre, device="cuda", time_sel=slice("2002-01-03", "2003-01-02"))
    host2 = af.dataset_from_path(pattern, "t2m", preprocess=pre, time_sel=slice("2002-01-03", "2003-01-02"))
    assert dev2.time.equals(host2.time)
    np.testing.assert_array_equal(dev2.cube().cpu().numpy(), host2.cube())


def test_clip_to_regions_reads_only_the_box(torch_cuda, tmp_path, monkeypatch):
    """georegions= on the streaming route: only the chunks touching the regions' extent are read and only the
    box reaches HBM; same dataset (data, coordinates, cell ids) as the host route, 0-360 and +-180 stores."""
    from aggfly_amd import codec
    T, ny, nx = 48, 40, 72
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=71, scattered_nan=9)
    time = pd.date_range("2005-07-01", periods=T, freq="h")
    lat = 20 + 1.0 * np.arange(ny)
    calls = []
    real = codec.decode_ranges
    monkeypatch.setattr(codec, "decode_ranges", lambda kind, locs, outs, threads=8: calls.append(len(locs)) or real(kind, locs, outs, threads))
    regions = af.GeoRegions(pd.DataFrame({"geoid": ["a", "b"], "minx": [-100.2, -95.0], "miny": [31.3, 35.0],
                                          "maxx": [-96.0, -90.4], "maxy": [38.0, 41.7]}))
    for lon, is360 in ((230 + 1.0 * np.arange(nx), True), (-130 + 1.0 * np.arange(nx), False)):
        ds = af.Dataset(_xr(cube, time, lat, lon), lon_is_360=is360)
        store = str(tmp_path / f"clip_{is360}.zarr")
        af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 24, "latitude": 8, "longitude": 12})
        host = af.dataset_from_path(store, "t2m", georegions=regions, lon_is_360=is360)
        calls.clear()
        dev = af.dataset_from_path(store, "t2m", georegions=regions, lon_is_360=is360, device="cuda")
        assert dev.cube().is_cuda and dev.cube().shape == host.cube().shape and dev.cube().shape[1] < ny and dev.cube().shape[2] < nx
        np.testing.assert_array_equal(dev.cube().cpu().numpy(), host.cube())
        np.testing.assert_array_equal(dev.latitude, host.latitude)
        np.testing.assert_array_equal(dev.longitude, host.longitude)
        np.testing.assert_array_equal(dev.grid.cell_id, host.grid.cell_id)
        assert 0 < sum(calls) < 2 * 5 * 6                                             # of 60 chunk files


def test_netcdf4_chunked_variable_streams_into_hbm(torch_cuda, monkeypatch):
    """A chunked netCDF-4 variable (shuffle + deflate [+ fletcher32] chunks, or unfiltered chunks) takes the same
    streaming route as a Zarr array: byte ranges of the .nc file inflated and unshuffled by the native codec,
    placed on the GPU.  Files written by the real HDF5 library (tests/golden/hdf5)."""
    from aggfly_amd import codec
    fix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")
    calls = []
    real = codec.decode_ranges
    monkeypatch.setattr(codec, "decode_ranges", lambda kind, locs, outs, threads=8: calls.append(kind) or real(kind, locs, outs, threads))
    rec = af.dataset_from_path(os.path.join(fix, "unlimited_time.nc"), "t2m", device="cuda")     # record dimension, 1-step chunks
    np.testing.assert_array_equal(rec.cube().cpu().numpy(), af.dataset_from_path(os.path.join(fix, "unlimited_time.nc"), "t2m").cube())
    for fn in ("nc4_like.nc", "old_style.h5"):
        for var in ("t2m", "t2m_chunked_nofilter", "t2m_packed"):                 # the packed one is contiguous: host read, one upload
            host = af.dataset_from_path(os.path.join(fix, fn), var)
            dev = af.dataset_from_path(os.path.join(fix, fn), var, device="cuda")
            assert dev.cube().is_cuda and dev.time.equals(host.time)
            np.testing.assert_array_equal(dev.cube().cpu().numpy(), host.cube())
            np.testing.assert_array_equal(dev.latitude, host.latitude)
        sel = af.dataset_from_path(os.path.join(fix, fn), "t2m", device="cuda", time_sel=slice("2000-01-03", "2000-01-05"))
        hsel = af.dataset_from_path(os.path.join(fix, fn), "t2m", time_sel=slice("2000-01-03", "2000-01-05"))
        assert sel.time.equals(hsel.time) and len(sel.time) == 12
        np.testing.assert_array_equal(sel.cube().cpu().numpy(), hsel.cube())
    assert ("zlib", 4) in calls and "raw" in calls                                 # the native route really ran


def test_packed_int16_store_streams_packed_and_unpacks_in_hbm(torch_cuda, tmp_path):
    """ERA5-style packing (int16 + scale_factor / add_offset / _FillValue): the streaming route moves the
    packed integers over PCIe and applies the CF decoding in HBM — bit-identical to the host route, for
    time-contiguous and space-tiled chunks, Blosc and raw, with an absent chunk."""
    from aggfly_amd import io as afio
    T, ny, nx = 24 * 20, 8, 12
    rng = np.random.default_rng(8)
    packed = rng.integers(-30000, 30000, (T, ny, nx)).astype(np.int16)
    packed[rng.random((T, ny, nx)) < 0.02] = -32767
    attrs = {"scale_factor": 0.0017, "add_offset": 281.3, "_FillValue": -32767}
    time = pd.date_range("2004-03-01", periods=T, freq="h")
    lat, lon = 35 + 0.25 * np.arange(ny), 250 + 0.25 * np.arange(nx)
    tv, tattrs = afio._encode_time(time)
    for comp, chunks in (({"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}, (48, ny, nx)),
                         (None, (48, ny, nx)), ({"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}, (100, 4, 5))):
        store = str(tmp_path / f"p_{bool(comp)}_{chunks[1]}.zarr")
        os.makedirs(store)
        json.dump({"zarr_format": 2}, open(os.path.join(store, ".zgroup"), "w"))
        afio._write_array(store, "t2m", packed, ("time", "latitude", "longitude"), chunks, attrs, comp)
        afio._write_array(store, "time", np.asarray(tv, dtype=np.float64), ("time",), (T,), tattrs, None)
        afio._write_array(store, "latitude", lat, ("latitude",), (ny,), {}, None)
        afio._write_array(store, "longitude", lon, ("longitude",), (nx,), {}, None)
        os.remove(afio.ZarrArray(os.path.join(store, "t2m")).chunk_path((1, 0, 0)))       # an absent chunk = fill value
        host = af.dataset_from_path(store, "t2m")
        dev = af.dataset_from_path(store, "t2m", device="cuda")
        assert dev.cube().is_cuda and dev.cube().dtype == torch_cuda.float32 and host.cube().dtype == np.float32
        np.testing.assert_array_equal(dev.cube().cpu().numpy(), host.cube())
        assert np.isnan(host.cube()).mean() > 0.01


def test_float32_reference_rounding_mode(torch_cuda):
    """On float32 cubes the reference's intermediates are float32 (each step stores its output in
    its input dtype, nb_kernels.py:257-262).  By default this engine keeps float64 (more accurate,
    ~1e-7 away from an all-float32 reference run); with match_reference_f32 it reproduces the
    reference's roundings and must agree with the oracle's numba path RUN ON THE FLOAT32 INPUT."""
    from aggfly_amd import engine as eng
    ds, w, ods64, ow = _mid_case(np.float32)
    ods32 = ra.ODataset(ds.cube().copy(), ods64.time, ods64.latitude, ods64.longitude, ods64.lon_is_360)
    spec = dict(C2_SPEC,
                mx=[("aggregate", {"calc": "max", "groupby": "date"}), ("aggregate", {"calc": "mean", "groupby": "month"})],
                sp=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "spline"}),
                    ("aggregate", {"calc": "sum", "groupby": "month"})])
    want32 = ra.aggregate_dataset(ow, ods32, engine="numba", **spec)
    cols = [c for c in want32.columns if c not in ("geoid", "time")]
    old = (eng.config.match_reference_f32, eng.config.exact_order)
    try:
        eng.config.match_reference_f32, eng.config.exact_order = True, True
        got = af.aggregate_dataset(dataset=ds, weights=w, **spec)
        np.testing.assert_allclose(got[cols].values, want32[cols].values, rtol=1e-12, atol=0, equal_nan=True)
        eng.config.match_reference_f32 = False
        plain = af.aggregate_dataset(dataset=ds, weights=w, **spec)
    finally:
        eng.config.match_reference_f32, eng.config.exact_order = old
    # the default (all-float64) results sit ~1e-7 from the float32 reference run, as documented
    rel = np.nanmax(np.abs(plain[cols].values - want32[cols].values) / np.maximum(np.abs(want32[cols].values), 1e-30))
    assert 1e-12 < rel < 1e-4


def test_concurrent_host_threads_share_cached_plans(torch_cuda):
    """The reference's kernels are called from a thread pool (`nb_kernels.py:271-305`); here eight host threads run the
    same spec on different cubes at once.  They share ONE cached plan (and its scratch in HBM): every result must equal
    the result of the same call made alone."""
    from concurrent.futures import ThreadPoolExecutor
    from aggfly_amd import synth
    T, ny, nx = 24 * 60, 20, 24
    time = pd.date_range("2001-01-01", periods=T, freq="h")
    lat, lon = 30 + 0.25 * np.arange(ny), 250 + 0.25 * np.arange(nx)
    tab = synth.weights_table(ny, nx, 7, seed=9)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
    spec = dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "month"})],
                t=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 4)}),
                   ("aggregate", {"calc": "sum", "groupby": "month"})])
    dss = []
    for i in range(8):
        cube = synth.temperature_cube(T, ny, nx, dtype=np.float64, seed=100 + i, ocean_frac=0.1, scattered_nan=5)
        dss.append(af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}),
                              lon_is_360=True).to_device())
    w = af.weights_from_objects(dss[0], gr, table=tab)
    alone = [af.aggregate_dataset(dataset=d, weights=w, **spec) for d in dss]
    for _ in range(3):
        with ThreadPoolExecutor(max_workers=8) as ex:
            together = list(ex.map(lambda d: af.aggregate_dataset(dataset=d, weights=w, **spec), dss))
        for a, b in zip(alone, together):
            pd.testing.assert_frame_equal(a, b, check_exact=True)
    assert not alone[0].drop(columns=["geoid", "time"]).equals(alone[1].drop(columns=["geoid", "time"]))     # the cubes do differ


def test_weekly_panel_takes_the_region_fused_period_ends(torch_cuda):
    """`aggregate_dataset` with many output periods on a grid large enough to stream in whole periods (13 weeks of hourly float64
    data on 160 x 256 cells, 30 regions): the cached plan reduces its cells by region inside the streaming kernel at every period
    end (afhip_kernels.h: rf_emit) — checked through the public API against the oracle's whole pipeline (temporal stage, power
    transform, shared validity, weighted means, long frame) at the 1e-10 of the other API tests, and from four host threads at
    once (they share the plan, the CSR handle and its lazily built run tables)."""
    from concurrent.futures import ThreadPoolExecutor
    from aggfly_amd import engine as eng
    T, ny, nx = 24 * 91, 160, 256
    time = pd.date_range("2001-01-01", periods=T, freq="h")           # a Monday: 13 whole weeks
    lat, lon = 20 + 0.25 * np.arange(ny), 230 + 0.25 * np.arange(nx)
    tab = synth.weights_table(ny, nx, 30, seed=19, zero_frac=0.05)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
    spec = dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "week"})],
                t=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                   ("aggregate", {"calc": "mean", "groupby": "week"})])
    cubes = [synth.temperature_cube(T, ny, nx, dtype=np.float64, seed=200 + i, ocean_frac=0.1, scattered_nan=20) for i in range(2)]
    dss = [af.Dataset(af.DataArray(c, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}), lon_is_360=True).to_device()
           for c in cubes]
    w = af.weights_from_objects(dss[0], gr, table=tab)
    eng._PLAN_CACHE.clear(); eng._CSR_CACHE.clear()
    got = [af.aggregate_dataset(dataset=d, weights=w, **spec) for d in dss]
    plans = list(eng._PLAN_CACHE.values())
    assert len(plans) == 1 and "last-run=region-fused" in plans[0].describe(), [p.describe() for p in plans]
    ow = ra.OWeights(tab, np.arange(ny * nx), gr.shp["geoid"], "geoid", "nan")
    for c, df in zip(cubes, got):
        want = ra.aggregate_dataset(ow, ra.ODataset(c, time, lat, lon, True), engine="numba", **spec)
        cols = [k for k in want.columns if k not in ("geoid", "time")]
        assert list(df.columns) == list(want.columns) and len(df) == len(want) and (df["time"].values == want["time"].values).all()
        np.testing.assert_allclose(df[cols].values, want[cols].values, rtol=1e-10, atol=1e-10, equal_nan=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        together = list(ex.map(lambda d: af.aggregate_dataset(dataset=d, weights=w, **spec), dss * 2))
    for a, b in zip(got * 2, together):
        pd.testing.assert_frame_equal(a, b, check_exact=True)


def test_handles_are_bound_to_the_cubes_device_from_a_fresh_thread(torch_cuda):
    """Handles belong to a device (include/aggfly_hip.h "Devices").  A fresh host thread starts on device 0 whatever card
    its parent selected — the threaded caller the reference supports (`nb_kernels.py:271-305` under dask's pool) would, on
    rank r > 0 of an 8-GPU node, otherwise build its tables on card 0 and launch on a cube that lives on card r.  Here the
    cube sits on the LAST visible device (device 0 on a one-GPU box), the main thread's current device stays 0, and the
    call is made from a new thread: the cached plan and CSR must report the cube's device, the result must equal the oracle,
    a cube of another device must be refused, and the calling thread's current device must come back unchanged."""
    import threading
    from aggfly_amd import engine as eng, hip, synth
    torch = torch_cuda
    dev = torch.cuda.device_count() - 1
    T, ny, nx = 24 * 40, 12, 16
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float64, seed=77, ocean_frac=0.1, scattered_nan=5)
    time = pd.date_range("2001-01-01", periods=T, freq="h")
    lat, lon = 30 + 0.25 * np.arange(ny), 250 + 0.25 * np.arange(nx)
    tab = synth.weights_table(ny, nx, 5, seed=9)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
    spec = dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "month"})],
                s=[("aggregate", {"calc": "sine_dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "month"})])
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}),
                    lon_is_360=True).to_device(f"cuda:{dev}")
    assert eng.dataset_device(ds) == dev
    w = af.weights_from_objects(ds, gr, table=tab)
    eng._PLAN_CACHE.clear(); eng._CSR_CACHE.clear()
    box = {}

    def job():
        box["before"] = torch.cuda.current_device()
        box["df"] = af.aggregate_dataset(dataset=ds, weights=w, **spec)
        box["after"] = torch.cuda.current_device()

    t = threading.Thread(target=job)
    t.start(); t.join()
    assert box["before"] == box["after"]                              # the library restores the caller's device
    lib = hip.load()
    plans, csrs = list(eng._PLAN_CACHE.items()), list(eng._CSR_CACHE.items())
    assert plans and csrs
    for key, plan in plans:
        assert key[0] == dev and plan.device_index == dev and lib.afhip_plan_device(plan._h) == dev
    for key, (csr, _) in csrs:
        assert key[0] == dev and csr.device_index == dev and lib.afhip_csr_device(csr.handle) == dev
    ow = ra.OWeights(tab, np.arange(ny * nx), gr.shp["geoid"], "geoid", "nan")
    want = ra.aggregate_dataset(ow, ra.ODataset(cube, time, lat, lon, True), engine="numba", **spec)
    cols = [c for c in want.columns if c not in ("geoid", "time")]
    assert list(box["df"].columns) == list(want.columns) and len(box["df"]) == len(want)
    np.testing.assert_allclose(box["df"][cols].values, want[cols].values, rtol=1e-10, atol=1e-10, equal_nan=True)
    if torch.cuda.device_count() >= 2:                                # a cube on another card than the plan: refused before any launch
        plan = plans[0][1]
        other = torch.zeros((plan.T, ny, nx), dtype=torch.float64, device=f"cuda:{(dev + 1) % torch.cuda.device_count()}")
        with pytest.raises(ValueError, match="lives on device"):
            plan.run_temporal(other)


def _reference_cache_feather(project_dir, table, module_dict, obj_dict):
    """Write ``table`` where the reference's ProjectCache would (`aggfly/cache/project_cache.py:46-47,207-226`):
    ``{project_dir}/tmp/GridWeights/mod-<sha>/<sha>.feather`` (+ the mod.yaml beside it), Feather V2 = the Arrow IPC file
    format, sha = sha256(json.dumps(str(dict)))[:15]."""
    import hashlib
    import pyarrow as pa
    sha = lambda d: hashlib.sha256(json.dumps(str(d), sort_keys=True).encode("utf8")).hexdigest()[:15]
    d = os.path.join(project_dir, "tmp", "GridWeights", f"mod-{sha(module_dict)}")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "mod.yaml"), "w").write("".join(f"{k}: {v}\n" for k, v in module_dict.items()))
    path = os.path.join(d, f"{sha(obj_dict)}.feather")
    with pa.OSFile(path, "wb") as f:
        t = pa.Table.from_pandas(table, preserve_index=False)
        with pa.ipc.new_file(f, t.schema) as w:
            w.write_table(t)
    return path


def test_feather_weights_cache_through_to_the_panel(torch_cuda, tmp_path, dataset_360, georegion):
    """N3: a weights table the reference cached as Feather V2 (`project_cache.py:72-100`, written by `grid_weights.py:169-196`
    with its extra columns) is read with `weights_from_feather` / found by the CLI in the project cache and carried through
    `aggregate_dataset` on the GPU: the G2 golden on the 0-360 dataset; then a mid-size 0-360 case with a cell that is
    absent from the climate grid (dropped, `spatial.py:171-172`) and a zero-weight region kept as a NaN row under
    zero_weight="nan" (`spatial.py:144-153`) against the oracle."""
    from aggfly_amd.cli import pipeline
    # ---- G2: the reference's own pinned table (test_aggregate.py:234-237), with the columns its cache file carries
    g2 = gi.g2_weights_table().sort_values("cell_id").reset_index(drop=True)
    g2["area_weight"] = g2["weight"] * 2.0
    g2["raster_weight"] = 0.5
    path = _reference_cache_feather(str(tmp_path / "proj"), g2, {"grid": "2x2", "regions": "hull"}, {"func": "weights", "raster_weights": "pop"})
    assert "/tmp/GridWeights/mod-" in path and path.endswith(".feather")
    assert pipeline.find_weights_table(SimpleNamespace(weights_table=None, project_dir=str(tmp_path / "proj"))) == path
    w = af.weights_from_feather(path, dataset_360, georegion)
    assert list(w.weights.columns[:3]) == ["cell_id", "index_right", "weight"] and w.zero_weight == "nan"
    df = af.aggregate_dataset(dataset=dataset_360.to_device(), weights=w, **gi.g2_spec())
    assert list(df.columns) == ["geoid", "time", "tavg_1", "tavg_2"]
    assert np.allclose(df[["tavg_1", "tavg_2"]].values, np.array(G["G2_panel"]["values"]))
    # ---- mid-size: 0-360 longitudes, an absent cell, a zero-weight region
    T, ny, nx = 24 * 40, 9, 14
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float64, seed=51, ocean_frac=0.1, scattered_nan=20)
    time = pd.date_range("2002-03-01", periods=T, freq="h")
    lat, lon = 30 + 0.5 * np.arange(ny), 170 + 1.5 * np.arange(nx)            # crosses 180: the +-180 re-sort reorders the columns
    tab = synth.weights_table(ny, nx, 7, seed=52, secondary=True)
    tab.loc[tab["index_right"] == 3, "weight"] = 0.0                              # a zero-weight region
    absent = pd.DataFrame({"cell_id": [ny * nx + 5, ny * nx + 9], "index_right": [1, 2], "weight": [0.7, 0.3]})
    tab = pd.concat([tab, absent], ignore_index=True)                             # cells the climate grid does not have
    tab["area_weight"] = tab["weight"]
    regions = pd.DataFrame({"geoid": [f"r{i}" for i in range(7)]})
    path2 = _reference_cache_feather(str(tmp_path / "proj2"), tab, {"grid": "9x14"}, {"func": "weights", "raster_weights": None})
    ds = af.Dataset(_xr(cube, time, lat, lon), lon_is_360=True)
    for policy in ("nan", "area"):
        w2 = af.weights_from_feather(path2, ds, af.GeoRegions(regions), zero_weight=policy)
        spec = dict(t=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                       ("aggregate", {"calc": "sum", "groupby": "month"})],
                    dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "month"})])
        got = af.aggregate_dataset(dataset=ds.deepcopy().to_device(), weights=w2, **spec)
        ow = ra.OWeights(tab, np.arange(ny * nx), regions["geoid"], "geoid", policy)
        want = ra.aggregate_dataset(ow, ra.ODataset(cube, time, lat, lon, True), engine="numba", **spec)
        assert list(got.columns) == list(want.columns) and len(got) == len(want)
        assert list(got["geoid"]) == list(want["geoid"])
        cols = ["t_1", "t_2", "dd"]
        np.testing.assert_allclose(got[cols].values, want[cols].values, rtol=1e-12, equal_nan=True)
        z = got[got["geoid"] == "r3"]
        assert (len(z) == 2 and z[cols].isna().all().all()) if policy == "nan" else len(z) == 0      # kept as NaN rows / dropped


@pytest.mark.parametrize("calendar", ["standard", "noleap"])
def test_mismatched_output_time_axes_follow_the_outer_join(torch_cuda, calendar):
    """Names with different output frequencies (`month` and `year`): the reference lays them on the UNION of their labels
    (`xr.combine_by_coords`, an outer join with NaN fill, spatial.py:90-97); a period that one name lacks is invalid for
    every name (shared validity, :114-119) and its rows are dropped (:144-153) — except the NaN rows of zero-weight
    regions under zero_weight="nan", which appear at every label of the union.  Against the oracle's restatement."""
    ny, nx = 8, 10
    ndays = 365 * 2 + 40
    if calendar == "standard":
        time = pd.date_range("2001-01-01", periods=ndays, freq="D")
        otime = time
    else:
        time = af.cf_range("2001-01-01", ndays, "D", "noleap")
        otime = cf_daily_index("noleap", ndays, start=(2001, 1, 1))
    cube = synth.temperature_cube(ndays, ny, nx, seed=81, steps_per_day=1, ocean_frac=0.1, scattered_nan=15)
    lat, lon = 30 + 0.5 * np.arange(ny), 250 + 0.5 * np.arange(nx)
    tab = synth.weights_table(ny, nx, 6, seed=82, secondary=True)
    tab.loc[tab["index_right"] == 2, "weight"] = 0.0
    regions = pd.DataFrame({"geoid": [f"r{i}" for i in range(6)]})
    spec = dict(m=[("aggregate", {"calc": "mean", "groupby": "month"})],
                y=[("aggregate", {"calc": "max", "groupby": "year"})],
                dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "year"})])
    ds = af.Dataset(_xr(cube, time, lat, lon), lon_is_360=True)
    for policy in ("nan", "area"):
        w = af.weights_from_objects(ds, af.GeoRegions(regions), table=tab, zero_weight=policy)
        got = af.aggregate_dataset(dataset=ds.deepcopy().to_device(), weights=w, **spec)
        ow = ra.OWeights(tab, np.arange(ny * nx), regions["geoid"], "geoid", policy)
        want = ra.aggregate_dataset(ow, ra.ODataset(cube, otime, lat, lon, True), engine="numba", **spec)
        assert list(got.columns) == list(want.columns) == ["geoid", "time", "m", "y", "dd"] and len(got) == len(want)
        ymd = lambda t: (t.year, t.month, t.day)
        assert [ymd(t) for t in got["time"]] == [ymd(t) for t in want["time"]] and list(got["geoid"]) == list(want["geoid"])
        np.testing.assert_allclose(got[["m", "y", "dd"]].values, want[["m", "y", "dd"]].values, rtol=1e-12, equal_nan=True)
        live = got[got["geoid"] != "r2"]
        # only the labels EVERY name has survive: the two year ends that are month ends of the data too (the third
        # year's label, 2003-12-31, has no monthly value: the data end in February 2003)
        assert sorted({str(t)[:10] for t in live["time"]}) == ["2001-12-31", "2002-12-31"]
        z = got[got["geoid"] == "r2"]           # union = 26 month ends + 2003-12-31
        assert (len(z) == 27 and z[["m", "y", "dd"]].isna().all().all()) if policy == "nan" else len(z) == 0
    # the eager pieces behave the same: aggregate_time then aggregate_space on names with different axes
    w = af.weights_from_objects(ds, af.GeoRegions(regions), table=tab, zero_weight="area")
    tdict = af.aggregate_time(dataset=ds.deepcopy().to_device(), weights=w, **spec)
    assert len(tdict["m"].time) == 26 and len(tdict["y"].time) == 3
    df = af.aggregate_space(tdict, w)
    assert sorted({str(t)[:10] for t in df["time"]}) == ["2001-12-31", "2002-12-31"]
