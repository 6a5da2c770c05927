"""Randomised spec fuzz: specs drawn from the DSL grammar (inner reducer x transform x outer
reducer x groupby pairs, multi-row ddargs, several names per call) on seeded data, the whole
public path against the oracle.  Catches lowering / variant-selection / chunk-merge bugs that
hand-written cases miss."""
import numpy as np
import pandas as pd
import pytest

import aggfly_amd as af
from aggfly_amd import synth
from oracle import ref_aggregate as ra

pytestmark = pytest.mark.gpu

STATS = ["mean", "sum", "min", "max", "nanmean"]
PAIRS = [("date", "month"), ("date", "year"), ("month", "year"), ("date", "week"), ("week", "month")]


def _dd(rng, multi):
    def row():
        lo = float(rng.choice([-5, 0, 5, 10, 12.5, 18]))
        return [lo, lo + float(rng.choice([5, 10, 20, 90])), int(rng.integers(0, 2))]
    return [row() for _ in range(int(rng.integers(2, 4)))] if multi else row()


def _random_steps(rng):
    g1, g2 = PAIRS[int(rng.integers(0, len(PAIRS)))]
    kind = int(rng.integers(0, 8))
    if kind == 0:      # single level
        calc = str(rng.choice(STATS + ["dd", "bins", "sine_dd"]))
        p = {"calc": calc, "groupby": str(rng.choice([g1, g2]))}
        if calc in ("dd", "bins", "sine_dd"):
            p["ddargs"] = _dd(rng, rng.random() < 0.4)
            if calc == "sine_dd":
                p["ddargs"] = [r[:2] + [r[2] % 2] for r in p["ddargs"]] if isinstance(p["ddargs"][0], list) else p["ddargs"]
        return [("aggregate", p)]
    inner = str(rng.choice(STATS + ["dd", "bins", "sine_dd"]))
    p1 = {"calc": inner, "groupby": g1}
    multi1 = False
    if inner in ("dd", "bins", "sine_dd"):
        multi1 = rng.random() < 0.3
        p1["ddargs"] = _dd(rng, multi1)
    steps = [("aggregate", p1)]
    fan = multi1
    if kind in (2, 3, 4) and not multi1:
        u = rng.random()
        if kind == 4:          # X2: product with a second array of the inner level's shape (filled in by the caller)
            steps.append(("transform", {"transform": "inter", "inter": "__OTHER__"}))
        elif u < 0.6:
            steps.append(("transform", {"transform": "power", "exp": np.arange(1, int(rng.integers(2, 5)))}))
            fan = True
        elif u < 0.75:         # non-integer and negative exponents: the kernel's pow() variants
            steps.append(("transform", {"transform": "power", "exp": np.array([0.5, float(rng.choice([1.5, 2.5, -1.0]))])}))
            fan = True
        else:
            steps.append(("transform", {"transform": "spline"}))
            fan = True
    outer = str(rng.choice(["sum", "mean", "min", "max", "dd", "bins", "nanmean"]))
    p2 = {"calc": outer, "groupby": g2}
    if outer in ("dd", "bins"):
        p2["ddargs"] = _dd(rng, (not fan) and rng.random() < 0.3)
    steps.append(("aggregate", p2))
    if kind == 7 and g2 == "month":
        steps.append(("aggregate", {"calc": "sum", "groupby": "year"}))
    return steps


def _time_axes(rng, T, spd, seed, whole_days=True):
    """(product index, oracle index, kept positions): standard or CF calendar, with a gap of whole
    days and a few single missing steps so that empty and ragged groups occur."""
    from oracle.ref_calendar import OracleCFIndex, cf_daily_index
    keep = np.ones(T, bool)
    if whole_days:
        g0 = int(rng.integers(spd * 5, T - spd * 12))
        keep[g0:g0 + spd * int(rng.integers(1, 5))] = False
        keep[rng.integers(0, T, 4)] = False
    else:       # single steps only, never a whole day: date groups of mixed lengths 1 .. spd, none empty
        miss = rng.integers(0, T, 9)
        miss = miss[np.unique(miss // spd, return_index=True)[1]]          # at most one missing step per day
        keep[miss] = False
    keep = np.nonzero(keep)[0]
    cal = [None, "noleap", "360_day"][seed % 3] if seed >= 16 else None
    step_h = 24 // spd
    if cal is None:
        t = pd.date_range("2001-11-17 00:00", periods=T, freq=f"{step_h}h")
        return t[keep], t[keep], keep
    prod = af.cf_range("2001-11-17", T, f"{step_h}h", cal)
    days = cf_daily_index(cal, (T + spd - 1) // spd, (2001, 11, 17))
    rep = np.repeat(np.arange(len(days)), spd)[:T]
    orc = OracleCFIndex(days.year[rep], days.month[rep], days.day[rep], (np.arange(T) % spd) * step_h, cal)
    return prod[keep], orc[keep], keep


@pytest.mark.parametrize("seed", list(range(44)) + [47, 53, 59, 65])      # (the last four: series with single steps missing)
def test_random_specs_match_oracle(torch_cuda, seed):
    rng = np.random.default_rng(1000 + seed)
    dtype = np.float64 if seed % 2 == 0 else np.float32
    spd = 24 if seed < 16 else int(rng.choice([1, 2, 24]))            # hourly; later seeds also daily / 12-hourly
    pair_seed = 28 <= seed < 36 or (seed >= 44 and seed % 3 == 0)      # (seeds beyond the parametrised ones: scripts/fuzz_more.py)
    quad_seed = 36 <= seed < 44 or (seed >= 44 and seed % 3 == 1)
    gaps_seed = seed >= 44 and seed % 6 == 5      # 12- / 8- / 6-hourly steps with single steps missing: date groups of mixed lengths -> the `_rag` form
    if pair_seed:
        spd = 2             # unbroken (tmin, tmax)-like pairs: the date groups are all two rows -> pair mode and its lean group ends
    if quad_seed:
        spd = 4             # unbroken 6-hourly steps: the date groups are all four rows -> the four-row form of the lean group end
    if gaps_seed:
        spd = int(rng.choice([2, 3, 4]))
    ndays = int(rng.integers(70, 130)) if spd == 24 else int(rng.integers(400, 800))
    T, ny, nx = spd * ndays + (int(rng.integers(0, 24)) if spd == 24 else 0), int(rng.integers(3, 9)), int(rng.integers(3, 12))
    cube = synth.temperature_cube(T, ny, nx, dtype=dtype, seed=seed, steps_per_day=spd, ocean_frac=0.1, scattered_nan=15)
    if seed < 16:
        time = otime = pd.date_range("2001-11-17 05:00", periods=T, freq="h")
    elif pair_seed or quad_seed:
        time = otime = pd.date_range("2001-11-17 00:00", periods=T, freq="12h" if pair_seed else "6h")
    else:
        time, otime, keep = _time_axes(rng, T, spd, seed, whole_days=not gaps_seed)
        cube = np.ascontiguousarray(cube[keep])
        T = len(keep)
    lon360 = bool(seed % 3 == 0)
    lat, lon = -10 + 0.5 * np.arange(ny), (170.0 if lon360 else -30.0) + 0.5 * np.arange(nx)
    tab = synth.weights_table(ny, nx, max(2, ny * nx // 9), seed=seed, secondary=bool(seed % 2), zero_frac=0.15)
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}), lon_is_360=lon360)
    w = af.weights_from_objects(ds, gr, table=tab)
    ow = ra.OWeights(tab, np.arange(ny * nx), gr.shp["geoid"], "geoid", "nan")
    ods = ra.ODataset(cube.astype(np.float64), otime, lat, lon, lon360)
    for trial in range(7):
        # all names of one call must share the output frequency (one panel time axis)
        out_freq, spec = None, {}
        for v in range(int(rng.integers(1, 4))):
            for _ in range(20):
                steps = _random_steps(rng)
                if not isinstance(time, pd.DatetimeIndex) and any(p.get("groupby") == "week" for k, p in steps if k == "aggregate"):
                    continue                                   # no calendar week on CF calendars (raises, tested elsewhere)
                last = [p["groupby"] for k, p in steps if k == "aggregate"][-1]
                if out_freq in (None, last):
                    out_freq = last
                    spec[f"v{v}"] = steps
                    break
        # an `inter` step multiplies by a random array of the inner level's shape, (time', lat, lon): the layout of a step
        # output in the oracle, in the reference's compiled engine, and on both the fused and the staged path here
        ospec = {}
        for name, steps in spec.items():
            osteps, psteps = [], []
            for k, prm in steps:
                if k == "transform" and prm.get("inter") == "__OTHER__":
                    inner = ra.OTemporalAggregator(**steps[0][1]).execute(ods)
                    other = rng.normal(1.0, 0.5, inner.values.shape)
                    other[rng.integers(0, other.shape[0]), rng.integers(0, ny), rng.integers(0, nx)] = np.nan
                    osteps.append((k, dict(prm, inter=other)))
                    psteps.append((k, dict(prm, inter=other.copy())))
                else:
                    osteps.append((k, prm))
                    psteps.append((k, prm))
            ospec[name], spec[name] = osteps, psteps
        with np.errstate(invalid="ignore", divide="ignore"):
            want = ra.aggregate_dataset(ow, ods, engine="numba", **ospec)    # names with different label sets: outer join, like the product
        got = af.aggregate_dataset(dataset=ds, weights=w, **spec)
        assert list(got.columns) == list(want.columns), spec
        assert len(got) == len(want), spec
        assert (got["geoid"].values == want["geoid"].values).all(), spec
        if isinstance(time, pd.DatetimeIndex):
            assert (got["time"].values == want["time"].values).all(), spec
        else:
            assert [(t.year, t.month, t.day) for t in got["time"]] == [(t.year, t.month, t.day) for t in want["time"]], spec
        cols = [c for c in got.columns if c not in ("geoid", "time")]
        assert got[cols].shape == want[cols].shape, (spec, list(got.columns), list(want.columns))
        np.testing.assert_allclose(got[cols].values.astype(float), want[cols].values.astype(float), rtol=1e-10, atol=1e-10,
                                   equal_nan=True, err_msg=repr(spec))


@pytest.mark.parametrize("seed", range(12))
def test_random_specs_on_the_region_fused_route(torch_cuda, seed):
    """The same kind of fuzz on shapes that reach the region-fused period ends (afhip_kernels.h: rf_emit): 3-hourly data with weekly
    or monthly panels on a 48 x 80 grid with a dozen regions — periods short enough to stay whole on a small grid, regions large enough
    for their runs to pay.  Random inner reducers (incl. threshold slots on float64), integer powers, outer sum / mean / min / max /
    dd, up to three names per call; the whole public path against the oracle, and the route really taken wherever the plan
    qualifies (float64, or float32 without threshold slots; at least two periods)."""
    from aggfly_amd import engine as eng
    rng = np.random.default_rng(7000 + seed)
    dtype = np.float64 if seed % 2 == 0 else np.float32
    spd, ndays, ny, nx = 8, int(rng.integers(60, 100)), 48, 80
    T = spd * ndays
    cube = synth.temperature_cube(T, ny, nx, dtype=dtype, seed=300 + seed, steps_per_day=spd, ocean_frac=0.1, scattered_nan=30)
    time = pd.date_range("2001-01-01 00:00", periods=T, freq="3h")
    lon360 = bool(seed % 3 == 0)
    lat, lon = 10 + 0.5 * np.arange(ny), (200.0 if lon360 else -60.0) + 0.5 * np.arange(nx)
    tab = synth.weights_table(ny, nx, 12, seed=seed, secondary=bool(seed % 2), zero_frac=0.1)
    if seed % 4 >= 2:
        # junctions of polygons: 3 % of the cells get one to three more regions each (cells in up to five regions: the route's "extras")
        nreg = int(tab.index_right.max()) + 1
        jc = np.repeat(rng.choice(ny * nx, ny * nx // 33, replace=False), 3)[: int(rng.integers(ny * nx // 33, 3 * (ny * nx // 33)))]
        more = pd.DataFrame({"cell_id": jc, "index_right": rng.integers(0, nreg, len(jc)), "weight": rng.uniform(0.05, 0.9, len(jc))})
        tab = (pd.concat([tab[["cell_id", "index_right", "weight"]], more]).drop_duplicates(["index_right", "cell_id"])
               .sort_values(["index_right", "cell_id"], kind="stable").reset_index(drop=True))
        assert tab.groupby("cell_id").size().max() >= 3
    gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}), lon_is_360=lon360).to_device()
    w = af.weights_from_objects(ds, gr, table=tab)
    ow = ra.OWeights(tab, np.arange(ny * nx), gr.shp["geoid"], "geoid", "nan")
    ods = ra.ODataset(cube.astype(np.float64), time, lat, lon, lon360)
    taken = 0
    for trial in range(4):
        g2 = ["week", "month"][trial % 2]          # (months of 240 steps are cut into several slots on this small grid: per-cell route)
        spec, thr = {}, False
        for v in range(int(rng.integers(1, 4))):
            # (eight rows per date group: the direct-load path, whose variants have region-fused twins)
            inner = str(rng.choice(["mean", "sum", "min", "max", "nanmean"] + (["dd", "bins"] if dtype == np.float64 else [])))
            p1 = {"calc": inner, "groupby": "date"}
            if inner in ("dd", "bins"):
                p1["ddargs"] = _dd(rng, False)
                thr = True
            steps = [("aggregate", p1)]
            if rng.random() < 0.5 and inner not in ("dd", "bins"):
                steps.append(("transform", {"transform": "power", "exp": np.arange(1, int(rng.integers(2, 4)))}))
            outer = str(rng.choice(["sum", "mean", "min", "max", "dd"]))
            p2 = {"calc": outer, "groupby": g2}
            if outer == "dd":
                p2["ddargs"] = _dd(rng, False)
            steps.append(("aggregate", p2))
            spec[f"v{v}"] = steps
        with np.errstate(invalid="ignore", divide="ignore"):
            want = ra.aggregate_dataset(ow, ods, engine="numba", **spec)
        eng._PLAN_CACHE.clear()
        got = af.aggregate_dataset(dataset=ds, weights=w, **spec)
        assert list(got.columns) == list(want.columns) and len(got) == len(want), spec
        assert (got["geoid"].values == want["geoid"].values).all() and (got["time"].values == want["time"].values).all(), spec
        cols = [c for c in got.columns if c not in ("geoid", "time")]
        np.testing.assert_allclose(got[cols].values.astype(float), want[cols].values.astype(float), rtol=1e-10, atol=1e-10,
                                   equal_nan=True, err_msg=repr(spec))
        descs = [p.describe() for p in eng._PLAN_CACHE.values()]
        assert not any("region-fused-capable" in d for d in descs), (descs, spec)      # a capable plan that ran took the route (this table allows it)
        taken += sum("last-run=region-fused" in d for d in descs)
    assert taken >= 1, "no trial of this seed reached the region-fused route"
