"""The oracle against every golden vector / known-answer test the reference holds for the
path, plus the oracle's own cross-checks (numba-path == dask-path,
C port == numpy restatement, sparse stage == pure loops).  CPU only."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_inputs as gi  # noqa: E402

from oracle import cport, ref_aggregate as ra, ref_temporal as rt
from oracle.ref_calendar import cf_daily_index
from oracle.ref_spatial import spatial_compute, wavg_loops

G = gi.goldens()


def _ds360():
    arr, time, lat, lon = gi.dataset_360_inputs()
    return ra.ODataset(arr, time, lat, lon, True)


def _w(zero_weight="nan"):
    return ra.OWeights(gi.g2_weights_table(), np.arange(4), pd.Series(["region_1"], index=[0]), "geoid", zero_weight)


@pytest.mark.parametrize("engine", ["numba", "dask"])
def test_G1_temporal_table(engine):
    out = ra.aggregate_time(_ds360(), gi.g1_spec(), engine=engine)
    assert list(out) == G["G1_temporal_table"]["columns"]
    tab = np.stack([out[n].values.reshape(-1) for n in out], axis=1)
    assert np.allclose(tab, np.array(G["G1_temporal_table"]["values"]))


@pytest.mark.parametrize("engine", ["numba", "dask"])
def test_G2_panel(engine):
    df = ra.aggregate_dataset(_w(), _ds360(), engine=engine, **gi.g2_spec())
    assert list(df.columns) == ["geoid", "time", "tavg_1", "tavg_2"]
    assert np.allclose(df[["tavg_1", "tavg_2"]].values, np.array(G["G2_panel"]["values"]))
    assert df["time"].iloc[0] == pd.Timestamp("2000-07-31")


def test_K10_deprecated_kwargs_warn_and_are_ignored():
    spec = dict(tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "month"})])
    ref = ra.aggregate_dataset(_w(), _ds360(), **spec)
    with pytest.warns(DeprecationWarning, match="no longer builds a Dask cluster"):
        got = ra.aggregate_dataset(_w(), _ds360(), n_workers=50, processes=True, cluster_args={}, **spec)
    assert "tavg" in got.columns and "n_workers" not in got.columns
    assert np.allclose(got["tavg"].values, ref["tavg"].values, equal_nan=True)


def test_K1_sine_dd_partial_nan_masking():
    time = pd.date_range("2000-07-01", periods=4, freq="12h")
    arr = np.empty((4, 2, 2))
    arr[0], arr[1], arr[2], arr[3] = 15.0, 30.0, 18.0, 28.0
    arr[1, 0, 1] = np.nan
    arr[0, 1, 0] = np.nan
    ds = ra.ODataset(arr, time, np.array([-45.0, 45.0]), np.array([10.0, 100.0]), False)
    spec = dict(cdd=[("aggregate", {"calc": "sine_dd", "groupby": "date", "ddargs": [20, 99, 0]})])
    outs = {e: ra.aggregate_time(ds, spec, engine=e)["cdd"].values for e in ("numba", "dask")}
    assert np.allclose(outs["numba"], outs["dask"], equal_nan=True)
    o = outs["dask"]                       # [day, lat, lon]
    assert np.isnan(o[0, 0, 1]) and np.isnan(o[0, 1, 0])
    assert np.isfinite(o[0, 0, 0]) and o[0, 0, 0] > 0
    assert np.isfinite(o[1, 0, 1]) and o[1, 0, 1] > 0


def test_K2_cftime_bounds():
    k = G["K2_bounds"]
    t360 = cf_daily_index("360_day", 720)
    b_m, lab_m = rt.resample_groups(t360, "ME")
    assert set(np.diff(b_m).tolist()) == {k["360_day_720_ME_group_size"]}
    assert len(lab_m) == k["360_day_720_ME_n_labels"] and lab_m[0].calendar == "360_day"
    b_y, _ = rt.resample_groups(t360, "YE")
    assert b_y.tolist() == k["360_day_720_YE_bounds"]
    b_nl, _ = rt.resample_groups(cf_daily_index("noleap", 365), "ME")
    assert np.diff(b_nl)[:3].tolist() == k["noleap_365_ME_first3"]


@pytest.mark.parametrize("calendar", ["360_day", "noleap"])
@pytest.mark.parametrize("nan", [False, True])
def test_K3_numba_dask_parity_on_cf_calendars(calendar, nan):
    arr, lat, lon = gi.cftime_cube(720, nan=nan)
    ds = ra.ODataset(arr, cf_daily_index(calendar, 720), lat, lon, False)
    for name, steps in gi.k3_specs().items():
        nb = ra.aggregate_time(ds, {"v": steps}, engine="numba")
        dk = ra.aggregate_time(ds, {"v": steps}, engine="dask")
        assert set(nb) == set(dk)
        for key in nb:
            assert np.allclose(nb[key].values, dk[key].values, rtol=1e-9, atol=1e-9, equal_nan=True), (name, key)


def test_K4_empty_interior_bin_is_nan():
    t = cf_daily_index("360_day", 90)
    keep = np.nonzero(t.month != 2)[0]
    arr = np.random.default_rng(1).normal(15, 10, (len(keep), 2, 2))
    ds = ra.ODataset(arr, t[keep], np.array([-45.0, 45.0]), np.array([10.0, 100.0]), False)
    spec = {"v": [("aggregate", {"calc": "mean", "groupby": "month"})]}
    a = ra.aggregate_time(ds, spec, engine="numba")["v"].values
    b = ra.aggregate_time(ds, spec, engine="dask")["v"].values
    assert a.shape[0] == b.shape[0] == 3
    assert np.all(np.isnan(a[1]))
    assert np.allclose(a, b, equal_nan=True)


def test_K5_end_to_end_360_day_keeps_calendar():
    arr = np.random.default_rng(3).normal(20, 15, (360, 2, 2))
    ds = ra.ODataset(arr, cf_daily_index("360_day", 360), np.array([-45.0, 45.0]), np.array([90.0, 270.0]), True)
    spec = dict(tavg=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "month"})])
    nb = ra.aggregate_dataset(_w(), ds, engine="numba", **spec)
    dk = ra.aggregate_dataset(_w(), ds, engine="dask", **spec)
    assert len(nb) == 12 and nb["time"].iloc[0].calendar == "360_day"
    assert np.allclose(nb["tavg"].values, dk["tavg"].values, equal_nan=True)


def test_K6_week_on_cf_calendar_raises():
    ds = ra.ODataset(np.random.rand(60, 2, 2), cf_daily_index("360_day", 60), np.array([-45.0, 45.0]), np.array([10.0, 100.0]), False)
    for engine in ("numba", "dask"):
        with pytest.raises(NotImplementedError, match="week"):
            ra.aggregate_time(ds, {"v": [("aggregate", {"calc": "mean", "groupby": "week"})]}, engine=engine)


@pytest.mark.parametrize("case", ["multiregion_nan", "dropna_empty_group"])
def test_K7_spatial_vs_pure_loops(case):
    vals, time, lat, lon, wdf = gi.k7_case(case)
    flat = {"v": vals.reshape(vals.shape[0], 4).T}
    loops = wavg_loops(flat, time.values, [0, 1, 2, 3], wdf, ["v"]).sort_values(["region_id", "time"]).reset_index(drop=True)
    got = spatial_compute(flat, time.values, wdf, np.arange(4)).sort_values(["region_id", "time"]).reset_index(drop=True)
    assert got.shape == loops.shape
    assert (got[["region_id", "time"]].values == loops[["region_id", "time"]].values).all()
    assert np.allclose(got["v"].values, loops["v"].values, equal_nan=True)


def test_K8_zero_weight_nan_policy_rows():
    # region 1 has zero total weight: kept as NaN under "nan", dropped under "area"/"drop" semantics of the frame
    wdf = pd.DataFrame({"cell_id": [0, 1, 2, 3], "index_right": [0, 0, 1, 1], "weight": [1.0, 1.0, 0.0, 0.0]})
    x = np.ones((4, 2)); x[:, 1] = np.nan            # whole second timestep missing
    t = pd.date_range("2000-01-01", periods=2).values
    nanp = spatial_compute({"tavg": x}, t, wdf, np.arange(4), zero_weight="nan")
    assert len(nanp[nanp.region_id == 0]) == 1 and len(nanp[nanp.region_id == 1]) == 2
    assert nanp[nanp.region_id == 1]["tavg"].isna().all()
    area = spatial_compute({"tavg": x}, t, wdf, np.arange(4), zero_weight="area")
    assert set(area.region_id) == {0}


def test_c_port_matches_numpy_restatement():
    rng = np.random.default_rng(11)
    for dtype in (np.float64, np.float32):
        cube = rng.normal(15, 12, (24 * 6 + 5, 5, 7)).astype(dtype)
        cube[rng.integers(0, cube.shape[0], 20), rng.integers(0, 5, 20), rng.integers(0, 7, 20)] = np.nan
        b = np.array(sorted(list(range(0, cube.shape[0], 24)) + [48, cube.shape[0]]), dtype=np.int64)
        for calc in rt.STAT_CODE:
            np.testing.assert_array_equal(cport.block_stat(cube, b, calc), rt.numba_stat(cube, b, rt.STAT_CODE[calc]))
        dda = [[10, 30, 0], [5, 18, 1]]
        np.testing.assert_array_equal(cport.block_dd(cube, b, dda), rt.numba_dd(cube, b, dda))
        np.testing.assert_array_equal(cport.block_bins(cube, b, dda), rt.numba_bins(cube, b, dda))
        np.testing.assert_allclose(cport.block_sine_dd(cube, b, dda), rt.numba_sine_dd(cube, b, dda),
                                   rtol=1e-13 if dtype == np.float64 else 1e-6, atol=1e-12 if dtype == np.float64 else 1e-5, equal_nan=True)
    blk = rng.normal(0, 1, (35, 9))
    ri, ci, w = rng.integers(0, 6, 50), rng.integers(0, 35, 50), rng.random(50)
    from oracle.ref_spatial import scatter_block
    np.testing.assert_array_equal(cport.scatter_block(blk, ri, ci, w, 6), scatter_block(blk, ri, ci, w, 6))


def test_T4_sine_parts_are_the_single_sine_integrals():
    """T4 has no reference-held numeric vector (the reference's sine tests are properties, `test_aggregate.py:382-427`), so
    the restatement of `_block_sine_dd`'s closed forms (`nb_kernels.py:224-249`) is anchored analytically instead: on a
    window whose mean is its mid-range the single-sine model is T(t) = tavg + alpha sin(2 pi t) over one day, and the
    cooling part must be the day's mean of max(T - thr, 0), the heating part the mean of max(thr - T, 0) — here by
    numerical quadrature, for thresholds below, inside and above the window.  (This pins the algebra of the formulas both the
    oracle and the kernel's `sine_arc` restate; their agreement with each other is the 1e-10 GPU test.)"""
    from scipy.integrate import quad
    rng = np.random.default_rng(77)
    for _ in range(40):
        tmin = float(rng.uniform(-10, 30))
        tmax = tmin + float(rng.uniform(0.5, 25))
        tavg, alpha = (tmin + tmax) / 2, (tmax - tmin) / 2
        T = lambda t: tavg + alpha * np.sin(2 * np.pi * t)
        for thr in (tmin - 3.0, tmin + 0.1 * (tmax - tmin), tavg, tmin + 0.93 * (tmax - tmin), tmax + 2.0):
            stats = [np.array([v]) for v in (tmin, tmax, tavg)]
            cool = float(rt._sine_part_cooling(thr, *stats)[0])
            heat = float(rt._sine_part_heating(thr, *stats)[0])
            # break points of the integrands: where the sine crosses the threshold
            pts = None
            if tmin < thr < tmax:
                th0 = np.arcsin((thr - tavg) / alpha) / (2 * np.pi)
                pts = sorted({(th0) % 1.0, (0.5 - th0) % 1.0})
            want_c = quad(lambda t: max(T(t) - thr, 0.0), 0, 1, points=pts, limit=200, epsabs=1e-12, epsrel=1e-12)[0]
            want_h = quad(lambda t: max(thr - T(t), 0.0), 0, 1, points=pts, limit=200, epsabs=1e-12, epsrel=1e-12)[0]
            assert abs(cool - want_c) <= 1e-9 * max(1.0, abs(want_c)), (tmin, tmax, thr, cool, want_c)
            assert abs(heat - want_h) <= 1e-9 * max(1.0, abs(want_h)), (tmin, tmax, thr, heat, want_h)
            assert abs((cool - heat) - (tavg - thr)) <= 1e-9 * max(1.0, abs(tavg - thr))        # max(x,0) - max(-x,0) = x, averaged


def _sine_fixture_errors(get_value):
    """Errors of an implementation over tests/golden/sine_dd_fixtures.json (exact values at 50 digits, mpmath:
    tests/golden/make_sine_fixtures.py).  ``get_value(case, ddargs_row) -> float``.  -> {class: (max abs, max rel where |v| > 1e-6,
    max abs / scale)}, scale = max(window range, |v|); NaN cases (|r| > 1 in the heating form) must be NaN."""
    import json
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sine_dd_fixtures.json")))
    dd = np.array(fx["ddargs"])
    worst = {"interior": [0.0, 0.0, 0.0], "near_edge": [0.0, 0.0, 0.0]}
    for c in fx["cases"]:
        got = get_value(c, dd[c["row"]])
        if c["value"] is None:
            assert np.isnan(got), c
            continue
        v, w = c["value_f64"], np.array(c["window"])
        ae = abs(got - v)
        t = worst["near_edge" if " * rng inside " in c["tag"] else "interior"]
        t[0] = max(t[0], ae)
        t[1] = max(t[1], ae / abs(v) if abs(v) > 1e-6 else 0.0)
        t[2] = max(t[2], ae / max(w.max() - w.min(), abs(v)))
    return worst, len(fx["cases"])


def test_T4_oracle_against_the_50_digit_sine_fixtures():
    """Both restatements of `_block_sine_dd` (`nb_kernels.py:202-251`: the plain-C port and the numpy one) against the exact
    single-sine integrals.  What the reference's own arithmetic achieves, so that the GPU test's bound can be read against it:
    away from the window's edges 2e-14 relative; with a threshold within 1e-3 ... 1e-9 of the range from an edge the closed
    forms lose digits in acos / sqrt(1 - r^2): up to ~4e-9 RELATIVE (on values of 1e-5 ... 1e-3), 3e-13 absolute, 3e-14 of
    the window's scale — i.e. the reference's libm path is itself only good to rtol 1e-8 there, in absolute terms excellent."""
    for name, fn in (("cport", lambda w, d: cport.block_sine_dd(w, np.array([0, len(w)]), d[None, :])[0, 0, 0, 0]),
                     ("numpy", lambda w, d: rt.numba_sine_dd(w, np.array([0, len(w)]), d[None, :])[0, 0, 0, 0])):
        worst, n = _sine_fixture_errors(lambda c, d: float(fn(np.array(c["window"])[:, None, None], d)))
        assert n > 2000
        assert worst["interior"][1] <= 2e-14 and worst["interior"][2] <= 5e-15, (name, worst)
        assert worst["near_edge"][0] <= 5e-13 and worst["near_edge"][2] <= 5e-14 and worst["near_edge"][1] <= 1e-8, (name, worst)
        assert worst["near_edge"][1] > 1e-10, (name, worst)       # the libm closed form is NOT good to 1e-10 relative near the edges
