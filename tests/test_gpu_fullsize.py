"""Parity at BASELINE.json's full sizes (configs[1]: one year hourly on the 215 x 1440
counties extent, fp64, 21.7 GB resident).  The oracle cannot run the whole cube in seconds,
so the checks are (a) the oracle on a random SAMPLE of cells and regions of the full-size
run, compared exactly, and (b) size-independent properties: constant fields, affine
linearity of the mean columns, the bins partition identity, split-vs-unsplit chunking."""
import numpy as np
import pytest

import pandas as pd

from aggfly_amd import hip, synth
from oracle import cport

pytestmark = pytest.mark.gpu

T, NY, NX = 8760, 215, 1440
C = NY * NX


def _cube(torch, seed=1, const=None):
    import bench
    if const is not None:
        return torch.full((T, NY, NX), float(const), dtype=torch.float64, device="cuda")
    return bench.make_cube(torch, T, NY, NX, torch.float64, seed)


@pytest.fixture(scope="module")
def setup(torch_cuda):
    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs ~50 GB of free HBM")
    cube = _cube(torch)
    # NaN "ocean" cells and scattered NaNs, applied on the device
    g = torch.Generator(device="cuda").manual_seed(3)
    ocean = torch.rand((NY, NX), generator=g, device="cuda") < 0.05
    cube[:, ocean] = float("nan")
    idx = torch.randint(0, T * C, (2000,), generator=g, device="cuda")
    cube.view(-1)[idx] = float("nan")
    tab = synth.weights_table(NY, NX, 3100, seed=7, secondary=True, zero_frac=0.02)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, C)
    ib = synth.hourly_bounds(T)
    ob = np.array([0, len(ib) - 1], dtype=np.int64)
    return dict(torch=torch, cube=cube, tab=tab, csr=csr, ib=ib, ob=ob, R=R)


def _c2_cols():
    cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    return cols + [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]


def test_full_size_sampled_cells_and_regions_match_oracle(setup):
    s = setup
    torch = s["torch"]
    plan = hip.FusedPlan(T, C, hip.F64, s["ib"], s["ob"], _c2_cols(), exact_order=True)
    out = plan.run(s["cube"], s["csr"], want_cells=True)
    cells = out["cells"]                                            # [K, 1, C]
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(C, 96, replace=False))
    sample = s["cube"].view(T, C)[:, torch.from_numpy(pick).cuda()].cpu().numpy().reshape(T, 1, -1)
    dd = cport.resample(cport.resample(sample, s["ib"], "dd", [10, 30, 0]), s["ob"], "sum").reshape(-1)
    got = cells[:, 0, torch.from_numpy(pick).cuda()].cpu().numpy()
    np.testing.assert_array_equal(got[0], dd)                       # bit-exact (same order, exact_order)
    m = cport.resample(sample, s["ib"], "mean")
    for e in (1, 2, 3, 4):
        want = cport.resample(np.power(m, e), s["ob"], "sum").reshape(-1)
        np.testing.assert_allclose(got[e], want, rtol=4e-16, atol=0, equal_nan=True)
    # regions: recompute sampled rows of the panel from the GPU's own cells, in table order
    cells_h = cells[:, 0, :].cpu().numpy()
    valid = ~np.isnan(cells_h).any(axis=0)
    tab = s["tab"]
    num = out["num"].cpu().numpy(); den = out["den"].cpu().numpy(); res = out["res"].cpu().numpy()
    for r in rng.choice(s["R"], 40, replace=False):
        sub = tab[tab["index_right"] == r]
        c, w = sub["cell_id"].to_numpy(), sub["weight"].to_numpy()
        d = 0.0
        for ci, wi in zip(c, w):
            d += wi * float(valid[ci])
        assert den[r, 0] == d
        for k in range(5):
            n = 0.0
            for ci, wi in zip(c, w):
                n += wi * (cells_h[k, ci] if valid[ci] else 0.0)
            assert num[k, r, 0] == n
            assert (np.isnan(res[k, r, 0]) and d == 0) or res[k, r, 0] == n / d
    # default (split) chunking agrees with the unsplit order to rounding
    out2 = hip.FusedPlan(T, C, hip.F64, s["ib"], s["ob"], _c2_cols()).run(s["cube"], s["csr"])
    np.testing.assert_allclose(out2["res"].cpu().numpy(), res, rtol=1e-12, equal_nan=True)


def test_full_size_constant_field_and_linearity(setup):
    s = setup
    torch = s["torch"]
    cols = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3)]
    plan = hip.FusedPlan(T, C, hip.F64, s["ib"], s["ob"], cols)
    const = _cube(torch, const=17.25)
    res = plan.run(const, s["csr"])["res"].cpu().numpy()
    den = plan.run(const, s["csr"])["den"].cpu().numpy()
    for k, e in enumerate((1, 2, 3)):
        ok = den[:, 0] != 0
        np.testing.assert_allclose(res[k, ok, 0], 365 * 17.25 ** e, rtol=1e-13)   # a weighted mean of a constant
        assert np.isnan(res[k, ~ok, 0]).all()
    del const
    # mean -> pow 1 -> sum is affine in the data: f(a x + b) = a f(x) + 365 b on every region
    base = plan.run(s["cube"], s["csr"])["res"][0].clone()
    s["cube"].mul_(1.5).add_(4.0)
    moved = plan.run(s["cube"], s["csr"])["res"][0]
    s["cube"].sub_(4.0).div_(1.5)
    np.testing.assert_allclose(moved.cpu().numpy(), 1.5 * base.cpu().numpy() + 365 * 4.0, rtol=1e-11, equal_nan=True)


def test_full_size_bins_partition_identity(setup):
    s = setup
    edges = [-99.0, 0.0, 10.0, 20.0, 30.0, 99.0]
    cols = [dict(inner="mean", outer="bins", outer_args=(edges[i], edges[i + 1], 0)) for i in range(5)]
    out = hip.FusedPlan(T, C, hip.F64, s["ib"], s["ob"], cols).run(s["cube"], s["csr"], want_cells=True)
    cells = out["cells"][:, 0, :]
    total = cells.sum(dim=0).cpu().numpy()
    # every cell's daily means fall in exactly one bin, except NaN days, which fall in none
    assert set(np.unique(total)).issubset(set(range(366)))
    assert (total == 365).mean() > 0.9
    res = out["res"].sum(dim=0).cpu().numpy()[:, 0]
    assert np.nanmax(res) <= 365 + 1e-9


# ---------------------------------------------------------------------------------------
# the other BASELINE.json configs at their full sizes: sampled cells against the oracle
# ---------------------------------------------------------------------------------------
def _fill(torch, T, ny, nx, dtype, seed, spd):
    g = torch.Generator(device="cuda").manual_seed(seed)
    cube = torch.empty((T, ny, nx), dtype=dtype, device="cuda")
    for k0 in range(0, T, 512):
        k1 = min(T, k0 + 512)
        k = torch.arange(k0, k1, device="cuda", dtype=torch.float32)
        base = 14.0 + 11.0 * torch.sin(2 * np.pi * torch.floor(k / spd) / 365.0)
        cube[k0:k1] = (base[:, None, None] + 6.0 * torch.randn((k1 - k0, ny, nx), generator=g, device="cuda", dtype=torch.float32)).to(dtype)
    idx = torch.randint(0, T * ny * nx, (500,), generator=g, device="cuda")
    cube.view(-1)[idx] = float("nan")
    return cube


def _sample_cells(torch, cube, n, seed):
    T = cube.shape[0]
    C = cube[0].numel()
    pick = np.sort(np.random.default_rng(seed).choice(C, n, replace=False))
    host = cube.view(T, C)[:, torch.from_numpy(pick).cuda()].cpu().numpy().astype(np.float64).reshape(T, 1, -1)
    return pick, host


def _need(torch, gb):
    free, _ = torch.cuda.mem_get_info()
    if free < gb * 1e9:
        pytest.skip(f"needs {gb} GB of free HBM")


def _check_sampled_regions(torch, out, tab, n_regions, n_pick, seed):
    """The spatial stage of a full-size run (`SpatialAggregator.compute`, spatial.py:103-133): for a sample of regions,
    recompute num / den / res on the host from the GPU's OWN per-cell output, entry by entry in table order — the order
    `np.add.at` visits them (spatial.py:185) — for all columns and periods at once.  Exact for exact_order plans."""
    cells = out["cells"]                                               # [K, P, C] in HBM
    num, den, res = (out[k].cpu().numpy() for k in ("num", "den", "res"))
    K, P = cells.shape[0], cells.shape[1]
    rows = tab["index_right"].to_numpy()
    order = np.argsort(rows, kind="stable")
    starts = np.searchsorted(rows[order], np.arange(n_regions + 1))
    cid, wts = tab["cell_id"].to_numpy()[order], tab["weight"].to_numpy()[order]
    checked = 0
    for r in np.random.default_rng(seed).choice(n_regions, n_pick, replace=False):
        c, w = cid[starts[r]:starts[r + 1]], wts[starts[r]:starts[r + 1]]
        vals = cells[:, :, torch.from_numpy(c).cuda()].cpu().numpy()  # [K, P, entries]
        valid = ~np.isnan(vals).any(axis=0)                            # [P, entries]: shared across the K columns
        d = np.zeros(P)
        n = np.zeros((K, P))
        for i in range(len(c)):
            d += w[i] * valid[:, i]
            n += w[i] * np.where(valid[:, i], vals[:, :, i], 0.0)
        np.testing.assert_array_equal(den[r], d)
        np.testing.assert_array_equal(num[:, r], n)
        with np.errstate(invalid="ignore", divide="ignore"):
            np.testing.assert_array_equal(res[:, r], np.where(d != 0, n / d, np.nan))
        checked += 1
    return checked


def test_c1_counties_extent_f32_storage(torch_cuda):
    """configs[0] on the counties extent as the production stores hold it: float32 (T = 8760, 215 x 1440), the reference's
    own plan mean@date -> power(1, 2) -> sum@year, 3,100 regions with area weights.  Sampled cells against the oracle run
    on the float64-cast input (the float64 contract), sampled regions in table order, and the
    reference's float32 dtype walk (match_reference_f32 roundings) against the oracle's numba path on the float32 input."""
    torch = torch_cuda
    _need(torch, 20)
    cube = _fill(torch, T, NY, NX, torch.float32, 41, 24)
    ib = synth.hourly_bounds(T)
    ob = np.array([0, len(ib) - 1], dtype=np.int64)
    cols = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2)]
    plan = hip.FusedPlan(T, C, hip.F32, ib, ob, cols, exact_order=True)
    tab = synth.weights_table(NY, NX, 3100, seed=43)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, C)
    out = plan.run(cube, csr, want_cells=True)
    pick, host = _sample_cells(torch, cube, 96, 42)                 # host: the sampled columns cast to float64
    got = out["cells"][:, 0, torch.from_numpy(pick).cuda()].cpu().numpy()
    m = cport.resample(host, ib, "mean")
    np.testing.assert_array_equal(got[0], cport.resample(m, ob, "sum").reshape(-1))       # power 1: exact
    np.testing.assert_array_equal(got[1], cport.resample(m * m, ob, "sum").reshape(-1))   # power 2 = x * x: exact
    assert _check_sampled_regions(torch, out, tab, R, 40, 44) == 40
    # the reference's own dtype walk on float32 data: inner mean stored as float32, np.power(float32, int64) -> float64
    rcols = [dict(c, rounding=hip.ROUND_INNER) for c in cols]
    got32 = hip.FusedPlan(T, C, hip.F32, ib, ob, rcols, exact_order=True).run_temporal(cube)[:, 0, torch.from_numpy(pick).cuda()].cpu().numpy()
    m32 = cport.resample(host, ib, "mean").astype(np.float32).astype(np.float64)
    for k, e in enumerate((1, 2)):
        np.testing.assert_allclose(got32[k], cport.resample(np.power(m32, e), ob, "sum").reshape(-1), rtol=4e-16, atol=0, equal_nan=True)
    fast = hip.FusedPlan(T, C, hip.F32, ib, ob, cols).run(cube, csr)
    np.testing.assert_allclose(fast["res"].cpu().numpy(), out["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)


def test_c3_forty_years_hourly_f32(torch_cuda):
    """configs[2]: ERA5 hourly, 40 years (T = 350,640) on the CONUS window, f32 storage, P = 40."""
    torch = torch_cuda
    _need(torch, 45)
    T, ny, nx = 350640, 104, 236
    cube = _fill(torch, T, ny, nx, torch.float32, 11, 24)
    ib = synth.hourly_bounds(T)
    years = pd.date_range("1981-01-01", periods=T, freq="h").year.values[ib[:-1]]
    ob = np.concatenate([[0], np.nonzero(np.diff(years))[0] + 1, [len(ib) - 1]]).astype(np.int64)
    assert len(ob) - 1 == 41 or len(ob) - 1 == 40
    cols = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2)] + \
           [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    plan = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, cols, exact_order=True)
    cells = plan.run_temporal(cube)
    pick, host = _sample_cells(torch, cube, 24, 12)
    got = cells[:, :, torch.from_numpy(pick).cuda()].cpu().numpy()
    m = cport.resample(host, ib, "mean")
    for k, e in enumerate((1, 2)):
        want = cport.resample(np.power(m, e), ob, "sum").reshape(len(ob) - 1, -1)
        np.testing.assert_allclose(got[k], want, rtol=4e-16, atol=0, equal_nan=True)
    want = cport.resample(cport.resample(host, ib, "dd", [10, 30, 0]), ob, "sum").reshape(len(ob) - 1, -1)
    np.testing.assert_array_equal(got[2], want)
    # the whole path: 3,100 regions with population (secondary) weights, 2 % of them zero-weight -> K x R x 40 panel
    tab = synth.weights_table(ny, nx, 3100, seed=13, secondary=True, zero_frac=0.02)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, ny * nx)
    out = plan.run(cube, csr, want_cells=True)
    assert torch.equal(torch.nan_to_num(out["cells"]), torch.nan_to_num(cells))
    assert _check_sampled_regions(torch, out, tab, R, 48, 14) == 48
    assert np.isnan(out["res"].cpu().numpy()).any(axis=(0, 2)).sum() >= int(0.02 * R)          # zero-weight regions: den 0 -> NaN
    fast = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, cols).run(cube, csr)                        # default order: to rounding
    np.testing.assert_allclose(fast["res"].cpu().numpy(), out["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)


def test_ref_global_quarter_degree_monthly_polynomial_f32(torch_cuda):
    """The reference's own published benchmark shape (`benchmarks/bench_engine.py:19-23,60-69`): one year of hourly float32 on the global
    0.25 degree grid (8760 x 721 x 1440, 36.4 GB), mean@date -> power[1..4] -> sum@month, 3,100 regions.  Sampled cells against the
    oracle on the float64-cast input, sampled regions in table order from the exact_order plan's own per-cell values, and the default
    plan — twelve time chunks, one per month, with the region-fused period ends (the per-cell monthly values are never written) —
    against the exact plan at 1e-12."""
    torch = torch_cuda
    _need(torch, 60)
    T, ny, nx = 8760, 721, 1440
    cube = _fill(torch, T, ny, nx, torch.float32, 51, 24)
    ib = synth.hourly_bounds(T)
    ob = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])]).astype(np.int64)
    cols = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    tab = synth.weights_table(ny, nx, 3100, seed=53, zero_frac=0.01)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, ny * nx)
    exact = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, cols, exact_order=True)
    out = exact.run(cube, csr, want_cells=True)
    pick, host = _sample_cells(torch, cube, 48, 52)
    got = out["cells"][:, :, torch.from_numpy(pick).cuda()].cpu().numpy()       # [K, 12, cells]
    m = cport.resample(host, ib, "mean")
    for k, e in enumerate((1, 2, 3, 4)):
        want = cport.resample(np.power(m, e), ob, "sum").reshape(12, -1)
        np.testing.assert_allclose(got[k], want, rtol=4e-16, atol=0, equal_nan=True)
    assert _check_sampled_regions(torch, out, tab, R, 32, 54) == 32
    del out
    fast_plan = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, cols)
    assert "chunks=12 " in fast_plan.describe() and "region-fused-capable" in fast_plan.describe(), fast_plan.describe()
    fast = fast_plan.run(cube, csr)
    assert "last-run=region-fused" in fast_plan.describe(), fast_plan.describe()
    want = exact.run(cube, csr)
    for key in ("num", "den", "res"):
        np.testing.assert_allclose(fast[key].cpu().numpy(), want[key].cpu().numpy(), rtol=1e-12, equal_nan=True, err_msg=key)


def test_c4_cmip6_daily_bins_251_years(torch_cuda):
    """configs[3]: daily tas 1850-2100 on a noleap calendar (T = 91,615), 180 x 288 cells, 13 bins/yr."""
    torch = torch_cuda
    _need(torch, 30)
    import aggfly_amd as af
    from aggfly_amd.timegroups import resample_groups
    T, ny, nx = 91615, 180, 288
    cube = _fill(torch, T, ny, nx, torch.float32, 21, 1)
    time = af.cf_range("1850-01-01", T, "D", "noleap")
    ob, labels = resample_groups(time, "YE")
    assert len(labels) == 251 and set(np.diff(ob).tolist()) == {365}
    edges = np.arange(-20, 50, 5.0)
    dda = [[edges[i], edges[i + 1], 0] for i in range(13)]
    plan = hip.FusedPlan(T, ny * nx, hip.F32, ob, np.arange(252), [dict(inner="bins", inner_args=r) for r in dda])
    assert "ibins_sl" in plan.describe()
    cells = plan.run_temporal(cube)
    pick, host = _sample_cells(torch, cube, 32, 22)
    want = cport.block_bins(host, ob, dda)[:, 0]                    # [G, cells, D]
    got = cells[:, :, torch.from_numpy(pick).cuda()].cpu().numpy()  # [D, G, cells]
    np.testing.assert_array_equal(np.transpose(got, (1, 2, 0)), want)
    assert (got.sum(axis=0) <= 365).all()
    # the whole path at full size: 3,600 regions x 251 years x 13 bins with cropland (secondary) weights.  The packed-count
    # gather (no per-cell output) must equal the panel route bit for bit, and sampled regions the table-order host sums.
    tab = synth.weights_table(ny, nx, 3600, seed=23, secondary=True, zero_frac=0.01)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, ny * nx)
    exact = hip.FusedPlan(T, ny * nx, hip.F32, ob, np.arange(252), [dict(inner="bins", inner_args=r) for r in dda], exact_order=True)
    assert "packed-counts16" in exact.describe()
    direct = exact.run(cube, csr, want_cells=False)
    via_panel = exact.run(cube, csr, want_cells=True)
    for key in ("num", "den", "res"):
        assert torch.equal(torch.nan_to_num(direct[key], nan=-1.0), torch.nan_to_num(via_panel[key], nan=-1.0)), key
    assert _check_sampled_regions(torch, via_panel, tab, R, 40, 24) == 40
    fast = plan.run(cube, csr, want_cells=False)                                                 # default plan: same panel to rounding
    np.testing.assert_allclose(fast["res"].cpu().numpy(), direct["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)
    # a table with four decades of row lengths (one region of 17,280 cells): segments + ordered combine
    skew = synth.weights_table(ny, nx, 3600, seed=25, skew="lognormal")
    Rs = int(skew["index_right"].max()) + 1
    csr_s = hip.CSR(skew["index_right"].to_numpy(), skew["cell_id"].to_numpy(), skew["weight"].to_numpy(), Rs, ny * nx)
    ex_s = exact.run(cube, csr_s, want_cells=True)
    assert _check_sampled_regions(torch, ex_s, skew, Rs, 40, 26) == 40
    np.testing.assert_allclose(plan.run(cube, csr_s)["res"].cpu().numpy(), ex_s["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)


def test_c5_sine_dd_tenth_degree_global(torch_cuda):
    """configs[4]: 0.1 deg global grid (1801 x 3600), (tmin, tmax) pairs per day, sine_dd -> annual sum."""
    torch = torch_cuda
    _need(torch, 30)
    T, ny, nx = 730, 1801, 3600
    cube = _fill(torch, T, ny, nx, torch.float32, 31, 2)
    ib = synth.hourly_bounds(T, 2)
    ob = np.array([0, 365], dtype=np.int64)
    plan = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, [dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum"),
                                                         dict(inner="sine_dd", inner_args=(0, 18, 1), outer="sum")], exact_order=True)
    cells = plan.run_temporal(cube)
    pick, host = _sample_cells(torch, cube, 48, 32)
    got = cells[:, 0, torch.from_numpy(pick).cuda()].cpu().numpy()
    for k, dd in enumerate(([10, 30, 0], [0, 18, 1])):
        want = cport.resample(cport.resample(host, ib, "sine_dd", dd), ob, "sum").reshape(-1)
        np.testing.assert_allclose(got[k], want, rtol=1e-10, atol=1e-9, equal_nan=True)
    # the same cube through the LEAN pair form of other columns (the daily mean of tmin / tmax and its polynomial, min, max):
    # sampled cells against the oracle — statistics bit-exact, powers within libm's last bit — and default chunking (the one period
    # cut over chunks) against the unsplit order
    lean_cols = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)] + \
                [dict(inner="min", outer="sum"), dict(inner="max", outer="mean")]
    lplan = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, lean_cols, exact_order=True)
    assert lplan.describe().split()[0].endswith("_pair_lean"), lplan.describe()
    lcells = lplan.run_temporal(cube)
    lgot = lcells[:, 0, torch.from_numpy(pick).cuda()].cpu().numpy()
    m = cport.resample(host, ib, "mean")
    for k, e in enumerate((1, 2, 3, 4)):
        np.testing.assert_allclose(lgot[k], cport.resample(np.power(m, e), ob, "sum").reshape(-1), rtol=4e-16 if e > 1 else 0, atol=0, equal_nan=True)
    np.testing.assert_array_equal(lgot[4], cport.resample(cport.resample(host, ib, "min"), ob, "sum").reshape(-1))
    np.testing.assert_array_equal(lgot[5], cport.resample(cport.resample(host, ib, "max"), ob, "mean").reshape(-1))
    lfree = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, lean_cols).run_temporal(cube)
    np.testing.assert_allclose(lfree.cpu().numpy(), lcells.cpu().numpy(), rtol=1e-12, equal_nan=True)
    del lcells, lfree, lplan
    # the whole path: 40,000 admin-2-like regions (7 M table entries), then the same on a log-normal table whose
    # largest region holds > 10^5 cells
    for kw, seed in ((dict(), 33), (dict(skew="lognormal"), 35)):
        tab = synth.weights_table(ny, nx, 40000, seed=seed, **kw)
        R = int(tab["index_right"].max()) + 1
        csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, ny * nx)
        out = plan.run(cube, csr, want_cells=True)
        assert _check_sampled_regions(torch, out, tab, R, 40, seed + 1) == 40
        if kw:      # the longest row as well: > 10^5 entries in one running sum
            big = int(tab.groupby("index_right").size().idxmax())
            assert (tab["index_right"] == big).sum() >= 100_000
            cells = out["cells"]
            sub = tab[tab["index_right"] == big]
            c, w = sub["cell_id"].to_numpy(), sub["weight"].to_numpy()
            vals = cells[:, 0, torch.from_numpy(c).cuda()].cpu().numpy()
            valid = ~np.isnan(vals).any(axis=0)
            d = 0.0
            n = np.zeros(2)
            for i in range(len(c)):
                d += w[i] * valid[i]
                n += w[i] * np.where(valid[i], vals[:, i], 0.0)
            assert out["den"][big, 0].item() == d and np.array_equal(out["num"][:, big, 0].cpu().numpy(), n)
        fast = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob, [dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum"),
                                                             dict(inner="sine_dd", inner_args=(0, 18, 1), outer="sum")]).run(cube, csr)
        np.testing.assert_allclose(fast["res"].cpu().numpy(), out["res"].cpu().numpy(), rtol=1e-11, equal_nan=True)
        del csr, out


def test_daily_panel_of_the_headline_columns_adds_up_to_the_annual_one(torch_cuda, monkeypatch):
    """bench.py's DAILY row at its full size (hourly f32 on 215 x 1440, 3,100 regions, P = 365, K = 5; weighted sums per region formed
    inside the streaming kernel at every period end): size-independent properties of a many-period panel —
      * the numerators of the 365 daily periods add up to the annual plan's (the outer level is a plain sum over the days), and every
        day's denominator is the year's (the NaN cells of this cube are the same every hour);
      * the region-fused route and the per-cell route (AFHIP_NO_REGION_FUSED=1) agree to rounding on the whole panel;
      * a sample of regions of the per-cell route's own output, entry by entry in table order (exact_order), like the other configs."""
    torch = torch_cuda
    _need(torch, 30)
    import bench
    cube = bench.make_cube(torch, T, NY, NX, torch.float32, 5)
    g = torch.Generator(device="cuda").manual_seed(6)
    cube[:, torch.rand((NY, NX), generator=g, device="cuda") < 0.05] = float("nan")          # "ocean": NaN at every step
    tab = synth.weights_table(NY, NX, 3100, seed=7, secondary=True, zero_frac=0.02)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, C)
    ib = synth.hourly_bounds(T)
    G1 = len(ib) - 1
    daily_cols = [dict(inner="dd", inner_args=(10, 30, 0))] + [dict(inner="mean", transform="pow", transform_arg=e) for e in (1, 2, 3, 4)]
    annual_cols = [dict(c, outer="sum") for c in daily_cols]
    daily = hip.FusedPlan(T, C, hip.F32, ib, np.arange(G1 + 1, dtype=np.int64), daily_cols)
    out = daily.run(cube, csr)
    assert "last-run=region-fused" in daily.describe(), daily.describe()
    num, den, res = (out[k].cpu().numpy() for k in ("num", "den", "res"))               # [K, R, 365], [R, 365]
    year = hip.FusedPlan(T, C, hip.F32, ib, np.array([0, G1], dtype=np.int64), annual_cols).run(cube, csr)
    np.testing.assert_allclose(num.sum(axis=2), year["num"].cpu().numpy()[:, :, 0], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(den, np.repeat(year["den"].cpu().numpy(), G1, axis=1), rtol=1e-13)
    monkeypatch.setenv("AFHIP_NO_REGION_FUSED", "1")
    plain = hip.FusedPlan(T, C, hip.F32, ib, np.arange(G1 + 1, dtype=np.int64), daily_cols)
    monkeypatch.delenv("AFHIP_NO_REGION_FUSED")
    pc = plain.run(cube, csr)
    assert "region-fused" not in plain.describe(), plain.describe()
    np.testing.assert_allclose(num, pc["num"].cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(res, pc["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(den, pc["den"].cpu().numpy(), rtol=1e-13)
    del out, pc
    exact = hip.FusedPlan(T, C, hip.F32, ib, np.arange(G1 + 1, dtype=np.int64), daily_cols, exact_order=True).run(cube, csr, want_cells=True)
    assert _check_sampled_regions(torch, exact, tab, R, 12, seed=8) == 12
    np.testing.assert_allclose(res, exact["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)
