"""GPU parity, kernel level: each C-ABI entry point against the oracle on seeded inputs.

Bars: bit-exact for sums/means/min/max/dd/bins (same per-cell operation order as the
reference's numba kernels, float64 accumulation, no FMA contraction); 1e-10 relative for
sine_dd (device acos/sin/atan/cos vs libm), as BASELINE.json's north_star states.
"""
import numpy as np
import pandas as pd
import pytest

from oracle import cport
from oracle.ref_spatial import scatter_block as ref_scatter, spatial_num_den
from aggfly_amd import synth

pytestmark = pytest.mark.gpu


def _cube(T, ny, nx, dtype, seed, nan=True):
    a = synth.temperature_cube(T, ny, nx, dtype=dtype, seed=seed, steps_per_day=24,
                               ocean_frac=0.1 if nan else 0.0, scattered_nan=50 if nan else 0)
    return a


def _bounds_with_gaps(T):
    # daily groups, one empty group and one ragged tail
    b = list(range(0, T, 24))
    b.insert(5, b[5])          # zero-width group
    b.append(T)
    return np.unique(np.array(b, dtype=np.int64), return_index=False) if False else np.array(sorted(b), dtype=np.int64)


SHAPES = [(24 * 9 + 7, 6, 20), (24 * 9 + 7, 5, 7)]   # rows 16-byte multiples / not (scalar fallback)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("calc", ["mean", "sum", "min", "max", "nanmean"])
def test_group_stat_bit_exact(torch_cuda, dtype, shape, calc):
    from aggfly_amd import hip
    T, ny, nx = shape
    cube = _cube(T, ny, nx, dtype, seed=1)
    bounds = _bounds_with_gaps(T)
    want = cport.block_stat(cube, bounds, calc)
    got = hip.group_stat(torch_cuda.from_numpy(cube).cuda(), bounds, calc).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_group_dd_bins_bit_exact(torch_cuda, dtype, shape):
    from aggfly_amd import hip
    T, ny, nx = shape
    cube = _cube(T, ny, nx, dtype, seed=2)
    bounds = _bounds_with_gaps(T)
    dda = [[10, 30, 0], [20, 99, 0], [-99, 12.5, 1]]
    d = torch_cuda.from_numpy(cube).cuda()
    np.testing.assert_array_equal(hip.group_dd(d, bounds, dda).cpu().numpy(), cport.block_dd(cube, bounds, dda))
    np.testing.assert_array_equal(hip.group_bins(d, bounds, dda).cpu().numpy(), cport.block_bins(cube, bounds, dda))
    # single threshold row keeps the trailing D axis (the host squeezes it, nb_kernels.py:303-304)
    one = hip.group_dd(d, bounds, [10, 30, 0]).cpu().numpy()
    np.testing.assert_array_equal(one[..., 0], cport.block_dd(cube, bounds, [10, 30, 0])[..., 0])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_group_sine_dd_1e10(torch_cuda, dtype):
    from aggfly_amd import hip
    T, ny, nx = 24 * 12, 6, 20
    cube = _cube(T, ny, nx, dtype, seed=3)
    bounds = _bounds_with_gaps(T)
    dda = [[10, 30, 0], [5, 18, 1]]
    want = cport.block_sine_dd(cube, bounds, dda)
    got = hip.group_sine_dd(torch_cuda.from_numpy(cube).cuda(), bounds, dda).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    tol = 1e-10 if dtype == np.float64 else 2e-6   # f32 output rounding
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol, equal_nan=True)


def test_many_thresholds_split_passes(torch_cuda):
    from aggfly_amd import hip
    T, ny, nx = 24 * 5, 4, 8
    cube = _cube(T, ny, nx, np.float64, seed=4)
    bounds = synth.hourly_bounds(T)
    edges = np.linspace(-20, 45, 21)
    dda = [[edges[i], edges[i + 1], 0] for i in range(20)]
    got = hip.group_bins(torch_cuda.from_numpy(cube).cuda(), bounds, dda).cpu().numpy()
    np.testing.assert_array_equal(got, cport.block_bins(cube, bounds, dda))


def test_scatter_block_table_order(torch_cuda):
    from aggfly_amd import hip
    ny, nx, nt = 12, 16, 37
    wdf = synth.weights_table(ny, nx, 9, seed=5, secondary=True)
    rng = np.random.default_rng(5)
    block = rng.normal(20, 5, (ny * nx, nt))
    ridx, cidx, w = wdf["index_right"].to_numpy(), wdf["cell_id"].to_numpy(), wdf["weight"].to_numpy()
    R = int(ridx.max()) + 1
    want = ref_scatter(block, ridx, cidx, w, R)
    csr = hip.CSR(ridx, cidx, w, R, ny * nx)
    got = csr.scatter_block(torch_cuda.from_numpy(block).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, want)      # same entry order, product rounded before the add


def test_spatial_wavg_shared_validity(torch_cuda):
    from aggfly_amd import hip
    ny, nx, nt, K = 10, 12, 5, 3
    wdf = synth.weights_table(ny, nx, 7, seed=6, zero_frac=0.2)
    rng = np.random.default_rng(6)
    x = rng.normal(20, 5, (K, ny * nx, nt))
    x[0, rng.integers(0, ny * nx, 30), rng.integers(0, nt, 30)] = np.nan
    x[2, rng.integers(0, ny * nx, 30), rng.integers(0, nt, 30)] = np.nan
    x[:, :8, 0] = np.nan                                     # a region wholly NaN at t0 -> den 0 -> NaN
    arrs = {f"v{k}": x[k] for k in range(K)}
    nums, den, region_ids = spatial_num_den(arrs, wdf, np.arange(ny * nx))
    R = len(region_ids)
    csr = hip.CSR(wdf["index_right"].to_numpy(), wdf["cell_id"].to_numpy(), wdf["weight"].to_numpy(), R, ny * nx)
    num_g, den_g, res_g = (t.cpu().numpy() for t in csr.wavg(torch_cuda.from_numpy(x).cuda()))
    np.testing.assert_array_equal(den_g, den)
    for k in range(K):
        np.testing.assert_array_equal(num_g[k], nums[f"v{k}"])
        with np.errstate(invalid="ignore", divide="ignore"):
            want = np.divide(nums[f"v{k}"], den, out=np.full_like(den, np.nan), where=den != 0)
        np.testing.assert_array_equal(res_g[k], want)


def _oracle_two_level(cube, ib, ob, cols):
    """Per-cell oracle for a fused plan: numba_resample -> transform -> numba_resample."""
    out = []
    for c in cols:
        a = cport.resample(cube, ib, c["inner"], c.get("inner_args"), False)
        tf = c.get("transform")
        if tf == "pow":
            a = np.power(a, c["transform_arg"])
        elif tf == "hinge":
            a = (a > c["transform_arg"]) * (a - c["transform_arg"])
        outer = c.get("outer", "identity")
        if outer != "identity":
            a = cport.resample(np.ascontiguousarray(a), ob, outer, c.get("outer_args"), False)
        out.append(a.reshape(a.shape[0], -1))
    return np.stack(out)      # [K, P, cells]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("exact", [True, False])
def test_fused_plan_c2_cells_and_panel(torch_cuda, dtype, exact):
    """configs[1]: dd[10,30] + poly(1..4), daily inner groups, annual sum — cell by cell."""
    from aggfly_amd import hip
    T, ny, nx = 24 * 40 + 5, 8, 24
    cube = _cube(T, ny, nx, dtype, seed=7)
    ib = synth.hourly_bounds(T)
    G1 = len(ib) - 1
    ob = np.array([0, 17, 17, G1], dtype=np.int64)        # 3 outer periods, the middle one empty
    cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    cols += [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    plan = hip.FusedPlan(T, ny * nx, hip.F64 if dtype == np.float64 else hip.F32, ib, ob, cols, exact_order=exact)
    wdf = synth.weights_table(ny, nx, 11, seed=8, secondary=True, zero_frac=0.1)
    R = int(wdf["index_right"].max()) + 1
    csr = hip.CSR(wdf["index_right"].to_numpy(), wdf["cell_id"].to_numpy(), wdf["weight"].to_numpy(), R, ny * nx)
    out = plan.run(torch_cuda.from_numpy(cube).cuda(), csr, want_cells=True)
    cells = out["cells"].cpu().numpy()
    want = _oracle_two_level(cube.astype(np.float64), ib, np.array([0, 17, 17, G1]), cols)
    assert np.array_equal(np.isnan(cells), np.isnan(want))
    if exact:
        # integer powers come from a double-double product chain: correctly rounded like libm
        np.testing.assert_allclose(cells, want, rtol=4e-16, atol=0, equal_nan=True)
        np.testing.assert_array_equal(cells[0], want[0])          # dd sums: same order, bit-exact
        np.testing.assert_array_equal(cells[1], want[1])          # mean -> pow 1 -> sum: bit-exact
    else:
        np.testing.assert_allclose(cells, want, rtol=1e-12, atol=1e-12, equal_nan=True)
    # panel vs the oracle's spatial stage on the oracle's cells
    arrs = {f"c{k}": want[k].T for k in range(len(cols))}
    nums, den, _ = spatial_num_den(arrs, wdf, np.arange(ny * nx))
    tol = dict(rtol=0, atol=0) if exact else dict(rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(out["den"].cpu().numpy(), den, **tol)
    for k in range(len(cols)):
        np.testing.assert_allclose(out["num"][k].cpu().numpy(), nums[f"c{k}"], rtol=1e-11, atol=1e-9)


def test_fused_plan_bins_of_daily_means_and_single_level(torch_cuda):
    from aggfly_amd import hip
    T, ny, nx = 24 * 30, 4, 12
    cube = _cube(T, ny, nx, np.float64, seed=9)
    ib = synth.hourly_bounds(T)
    ob = np.array([0, 10, 30], dtype=np.int64)
    cols = [dict(inner="mean", outer="bins", outer_args=(-99, 20, 0)),
            dict(inner="mean", outer="bins", outer_args=(20, 99, 0)),
            dict(inner="max", outer="mean"), dict(inner="min", outer="min"),
            dict(inner="nanmean", outer="max"), dict(inner="mean", transform="hinge", transform_arg=20.0, outer="sum")]
    plan = hip.FusedPlan(T, ny * nx, hip.F64, ib, ob, cols, exact_order=True)
    cells = plan.run_temporal(torch_cuda.from_numpy(cube).cuda()).cpu().numpy()
    want = _oracle_two_level(cube, ib, ob, cols)
    np.testing.assert_array_equal(cells, want)
    # single level: monthly sine_dd / dd / bins / max straight on the raw steps
    ib2 = np.array([0, 24 * 10, 24 * 10, T], dtype=np.int64)
    cols2 = [dict(inner="sine_dd", inner_args=(10, 30, 0)), dict(inner="dd", inner_args=(10, 30, 0)),
             dict(inner="bins", inner_args=(0, 15, 0)), dict(inner="max")]
    plan2 = hip.FusedPlan(T, ny * nx, hip.F64, ib2, np.arange(4), cols2)
    got2 = plan2.run_temporal(torch_cuda.from_numpy(cube).cuda()).cpu().numpy()
    want2 = _oracle_two_level(cube, ib2, None, cols2)
    np.testing.assert_array_equal(got2[1:], want2[1:])
    np.testing.assert_allclose(got2[0], want2[0], rtol=1e-10, atol=1e-10, equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_single_level_plan_panel_matches_the_definition(torch_cuda, dtype):
    """Single-level plan (every inner group is an output period) through combine + CSR + divide: same
    num / den / res with and without the per-cell output, and equal to the definition — NaN cells void
    a whole (cell, period) for every column, an empty period has no weight, float32 final rounding."""
    from aggfly_amd import hip
    T, ny, nx, R = 24 * 40, 5, 24, 7
    rng = np.random.default_rng(21)
    cube = _cube(T, ny, nx, dtype, seed=13)
    cube[rng.integers(0, T, 40), rng.integers(0, ny, 40), rng.integers(0, nx, 40)] = np.nan     # sparse NaN steps
    cube[:, 1, 3] = np.nan                                                                       # an "ocean" cell
    ib = np.array([0, 24 * 10, 24 * 10, 24 * 25, T], dtype=np.int64)                             # 4 periods, one empty
    cols = [dict(inner="mean", rounding=hip.ROUND_FINAL if dtype == np.float32 else 0), dict(inner="dd", inner_args=(10, 30, 0)),
            dict(inner="bins", inner_args=(0, 15, 0)), dict(inner="nanmean"), dict(inner="max")]
    C_ = ny * nx
    rows = rng.integers(0, R, 3 * C_ // 2); ccols = rng.integers(0, C_, 3 * C_ // 2); w = rng.uniform(0.1, 1.0, 3 * C_ // 2)
    order = np.argsort(rows, kind="stable")
    csr = hip.CSR(rows[order], ccols[order], w[order], R, C_)
    code = hip.F64 if dtype == np.float64 else hip.F32
    d = torch_cuda.from_numpy(cube).cuda()
    plan = hip.FusedPlan(T, C_, code, ib, np.arange(5), cols)
    assert "_sl" in plan.describe(), plan.describe()
    direct = plan.run(d, csr)
    twopass = plan.run(d, csr, want_cells=True)
    # (without the per-cell output the weighted sums gather the slots directly, lanes striding over a row's entries + a fixed
    # butterfly — k_csr_spmm_slots; with it they run in table order over the panel: the same numbers to rounding)
    for k in ("num", "den", "res"):
        np.testing.assert_allclose(direct[k].cpu().numpy(), twopass[k].cpu().numpy(), rtol=1e-13, atol=1e-12 if k == "num" else 0,
                                   equal_nan=True, err_msg=k)
    # and against the definition: shared validity, weighted sums in table order
    cells = twopass["cells"].cpu().numpy()                        # [K, P, C]
    valid = ~np.isnan(cells).any(axis=0)                          # [P, C]
    den = np.zeros((R, 4)); num = np.zeros((len(cols), R, 4))
    for r_, c_, w_ in zip(rows[order], ccols[order], w[order]):
        den[r_] += w_ * valid[:, c_]
        for k in range(len(cols)):
            num[k, r_] += w_ * np.where(valid[:, c_], cells[k, :, c_], 0.0)
    np.testing.assert_allclose(direct["den"].cpu().numpy(), den, rtol=1e-13, atol=0)
    np.testing.assert_allclose(direct["num"].cpu().numpy(), num, rtol=1e-12, atol=1e-12)
    assert (den[:, 1] == 0).all() and np.isnan(direct["res"].cpu().numpy()[:, :, 1]).all()      # the empty period


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_every_load_path_arm_gives_identical_cells(torch_cuda, dtype):
    """Direct loads (1/2/4 cells per lane, 4/8 rows in flight) and the LDS-DMA ring (depth 4/8/16,
    nt and default cache policy) are the same arithmetic: cells must match bit for bit."""
    from aggfly_amd import hip
    T, ny, nx = 24 * 33 + 11, 12, 44          # rows are 16-byte multiples for both dtypes
    cube = _cube(T, ny, nx, dtype, seed=13)
    ib = synth.hourly_bounds(T)
    ob = np.array([0, 9, 20, len(ib) - 1], dtype=np.int64)
    cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    cols += [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    code = hip.F64 if dtype == np.float64 else hip.F32
    d = torch_cuda.from_numpy(cube).cuda()
    ref = hip.FusedPlan(T, ny * nx, code, ib, ob, cols, exact_order=True).run_temporal(d).cpu().numpy()
    want = _oracle_two_level(cube.astype(np.float64), ib, ob, cols)
    np.testing.assert_array_equal(ref[:2], want[:2])
    arms = [104, 108, 204, 208, 1204, 1208, 1216, 11204] if dtype == np.float64 else [104, 108, 204, 208, 404, 408, 1404, 1408, 1416, 11404]
    seen = set()
    for arm in arms:
        plan = hip.FusedPlan(T, ny * nx, code, ib, ob, cols, exact_order=True, tuning=arm)
        seen.add(plan.describe().split()[0])
        np.testing.assert_array_equal(plan.run_temporal(d).cpu().numpy(), ref, err_msg=f"arm {arm}")
    # a `tuning` code is a hint: the production menu (`make`) holds the load paths the planner picks itself — direct loads with one
    # or two cells per lane and the LDS-DMA ring —, `make MENU=arms` every arm named above (hip.build_info() tells which build this is)
    if hip.build_info()["arms"] > 0:
        assert len(seen) == len(arms), seen          # every arm resolved to its own kernel
    else:
        assert len(seen) >= 2, seen


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ragged_group_lengths_on_every_arm(torch_cuda, dtype):
    """Inner groups of every length 0..19 in random order (shorter than, equal to and just past the 4 / 8 / 16
    rows a burst holds; the short-group rule picks the LDS ring by itself for some of them): bursts, batched
    tails and the ring give the same cells, and those equal the oracle."""
    from aggfly_amd import hip
    rng = np.random.default_rng(31)
    lens = np.concatenate([rng.permutation(20), rng.integers(0, 20, 60), [1, 1, 2, 3, 0, 0, 17]])
    ib = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T, ny, nx = int(ib[-1]), 6, 44
    cube = _cube(T, ny, nx, dtype, seed=17)
    G = len(lens)
    ob = np.array([0, 7, 7, 30, G], dtype=np.int64)                      # an empty outer period too
    cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum"), dict(inner="mean", transform="pow", transform_arg=2, outer="mean"),
            dict(inner="max", outer="max"), dict(inner="nanmean", outer="min"), dict(inner="bins", inner_args=(5, 25, 0), outer="sum")]
    code = hip.F64 if dtype == np.float64 else hip.F32
    d = torch_cuda.from_numpy(cube).cuda()
    want = _oracle_two_level(cube.astype(np.float64), ib, ob, cols)
    arms = [0, 104, 108, 204, 208, 1204, 1208] if dtype == np.float64 else [0, 104, 108, 204, 208, 404, 408, 1404, 1408]
    for arm in arms:
        got = hip.FusedPlan(T, ny * nx, code, ib, ob, cols, exact_order=True, tuning=arm).run_temporal(d).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"arm {arm}")
    # single level on the same ragged groups (sine_dd rides on min / max / mean of 0..19-step windows)
    cols1 = [dict(inner="sum"), dict(inner="min"), dict(inner="sine_dd", inner_args=(10, 30, 0)), dict(inner="dd", inner_args=(12, 28, 1))]
    want1 = _oracle_two_level(cube.astype(np.float64), ib, None, cols1)
    for arm in arms[:5]:
        got1 = hip.FusedPlan(T, ny * nx, code, ib, np.arange(G + 1), cols1, tuning=arm).run_temporal(d).cpu().numpy()
        np.testing.assert_array_equal(got1[[0, 1, 3]], want1[[0, 1, 3]], err_msg=f"arm {arm}")
        np.testing.assert_allclose(got1[2], want1[2], rtol=1e-10, atol=1e-10, equal_nan=True)


def test_edge_shapes(torch_cuda):
    """Ragged and degenerate inputs: odd row length (scalar-load fallback), one time step, groups
    shorter than the prefetch depth, an all-NaN cube, a weights table with a region whose cells
    are all absent and one with no entries at all."""
    from aggfly_amd import hip
    rng = np.random.default_rng(3)
    for (T, ny, nx) in [(1, 3, 5), (7, 1, 1), (50, 7, 9)]:
        cube = rng.normal(15, 10, (T, ny, nx))
        b = np.array(sorted(set([0, T] + list(range(0, T, 3)))), dtype=np.int64)
        for dt in (np.float64, np.float32):
            c = cube.astype(dt)
            d = torch_cuda.from_numpy(c).cuda()
            np.testing.assert_array_equal(hip.group_stat(d, b, "mean").cpu().numpy(), cport.block_stat(c, b, "mean"))
            np.testing.assert_array_equal(hip.group_bins(d, b, [[0, 15, 0], [15, 99, 0]]).cpu().numpy(),
                                          cport.block_bins(c, b, [[0, 15, 0], [15, 99, 0]]))
    # all-NaN cube: every statistic NaN, bins zero, panel rows all NaN
    T, ny, nx = 48, 4, 6
    nan_cube = np.full((T, ny, nx), np.nan)
    d = torch_cuda.from_numpy(nan_cube).cuda()
    ib = synth.hourly_bounds(T)
    assert np.isnan(hip.group_stat(d, ib, "sum").cpu().numpy()).all()
    assert (hip.group_bins(d, ib, [0, 15, 0]).cpu().numpy() == 0).all()
    # CSR with an empty region (row 1 has no entries) and an out-of-grid cell dropped by the host
    csr = hip.CSR([0, 0, 2], [0, 5, 23], [0.5, 0.5, 1.0], 3, ny * nx)
    plan = hip.FusedPlan(T, ny * nx, hip.F64, ib, np.array([0, 2]), [dict(inner="mean", outer="sum")])
    good = torch_cuda.from_numpy(rng.normal(10, 1, (T, ny, nx))).cuda()
    out = plan.run(good, csr)
    den = out["den"].cpu().numpy()[:, 0]
    res = out["res"].cpu().numpy()[0, :, 0]
    assert den.tolist() == [1.0, 0.0, 1.0] and np.isnan(res[1]) and np.isfinite(res[[0, 2]]).all()
    with pytest.raises(ValueError):
        hip.CSR([0], [ny * nx], [1.0], 1, ny * nx)            # column out of range is refused before upload
    with pytest.raises(ValueError):
        hip.FusedPlan(T, ny * nx, hip.F64, np.array([0, 30, 20, T]), np.array([0, 3]), [dict(inner="mean")])   # non-monotone bounds


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bins_partition_uses_lds_histogram_and_is_exact_on_edges(torch_cuda, monkeypatch, dtype):
    """Contiguous equal-width bins take the per-lane LDS-histogram path.  Strict compares must
    survive it: values exactly ON an edge fall in no bin, values one ulp to either side fall in
    the neighbouring bins, NaN/inf fall nowhere — exactly as the reference's compare chain."""
    from aggfly_amd import hip
    T, ny, nx = 365 * 2, 6, 10
    rng = np.random.default_rng(77)
    cube = rng.normal(14, 12, (T, ny, nx)).astype(dtype)
    edges = np.arange(-20, 50, 5.0)
    flat = cube.reshape(-1)
    pick = rng.choice(flat.size, 600, replace=False)
    e = rng.choice(edges, 600).astype(dtype)
    flat[pick[:200]] = e[:200]                                           # exactly on an edge
    flat[pick[200:400]] = np.nextafter(e[200:400], dtype(np.inf))        # one ulp above
    flat[pick[400:560]] = np.nextafter(e[400:560], dtype(-np.inf))       # one ulp below
    flat[pick[560:580]] = np.nan
    flat[pick[580:590]] = np.inf
    flat[pick[590:600]] = -np.inf
    bounds = np.array([0, 365, 365, T], dtype=np.int64)                  # an empty group in the middle
    dda = [[edges[i], edges[i + 1], 0] for i in range(13)]
    d = torch_cuda.from_numpy(cube).cuda()
    want = cport.block_bins(cube, bounds, dda)
    np.testing.assert_array_equal(hip.group_bins(d, bounds, dda).cpu().numpy(), want)
    cols = [dict(inner="bins", inner_args=r) for r in dda]
    code = hip.F64 if dtype == np.float64 else hip.F32
    # exact_order: the spatial sums below run in table order on every route (bit for bit against the host loop)
    plan = hip.FusedPlan(T, ny * nx, code, bounds, np.arange(4), cols, exact_order=True)
    assert "_hist" in plan.describe(), plan.describe()
    got = plan.run_temporal(d).cpu().numpy()                             # [D, G, cells]
    np.testing.assert_array_equal(np.transpose(got, (1, 2, 0)), want.reshape(3, -1, 13))
    # the whole pass: count plans leave the streaming kernel as packed 16-bit counts; without a per-cell output the
    # weighted sums gather them directly, with one the cell-major panel is built first — same numbers either way,
    # and both equal the reference's scatter of where(valid, x, 0) (empty middle group -> NaN panel column)
    assert "packed-counts" in plan.describe(), plan.describe()
    nreg = 7
    wr = rng.integers(0, nreg, 150); wc = rng.integers(0, ny * nx, 150); ww = rng.uniform(0.1, 2.0, 150)
    order = np.argsort(wr, kind="stable")
    csr = hip.CSR(wr[order], wc[order], ww[order], nreg, ny * nx)
    direct = plan.run(d, csr, want_cells=False)
    via_panel = plan.run(d, csr, want_cells=True)
    for key in ("num", "den", "res"):
        np.testing.assert_array_equal(direct[key].cpu().numpy(), via_panel[key].cpu().numpy(), err_msg=key)
    cells = np.transpose(got, (1, 2, 0))                                 # [G, cells, D], exact counts (NaN for the empty group)
    for g in range(3):
        valid = ~np.isnan(cells[g]).any(axis=1)
        den = np.zeros(nreg); num = np.zeros((13, nreg))
        for r_, c_, w_ in zip(wr[order], wc[order], ww[order]):
            den[r_] += w_ * valid[c_]
            for k in range(13):
                num[k, r_] += w_ * (cells[g, c_, k] if valid[c_] else 0.0)
        np.testing.assert_array_equal(direct["den"].cpu().numpy()[:, g], den)
        np.testing.assert_array_equal(direct["num"].cpu().numpy()[:, :, g], num)
    # 13 counts of up to 365 fit a 16-byte record (9-bit fields); one 730-step period needs 10 bits -> 16-bit fields, 32 bytes
    assert "packed-counts16" in plan.describe(), plan.describe()
    whole = np.array([0, T], dtype=np.int64)
    plan_w = hip.FusedPlan(T, ny * nx, code, whole, np.arange(2), cols, exact_order=True)
    assert "packed-counts32" in plan_w.describe(), plan_w.describe()
    want_w = cport.block_bins(cube, whole, dda).reshape(1, -1, 13)
    np.testing.assert_array_equal(np.transpose(plan_w.run_temporal(d).cpu().numpy(), (1, 2, 0)), want_w)
    dw, pw = plan_w.run(d, csr, want_cells=False), plan_w.run(d, csr, want_cells=True)
    for key in ("num", "den", "res"):
        np.testing.assert_array_equal(dw[key].cpu().numpy(), pw[key].cpu().numpy(), err_msg=key)
    np.testing.assert_array_equal(np.transpose(pw["cells"].cpu().numpy(), (1, 2, 0)), want_w)
    # 32-byte records through the TILED panel kernel (>= 4 periods) and the direct gather: four periods of 1,200 steps
    T4 = 4800
    cube4 = rng.normal(14, 12, (T4, ny, nx)).astype(dtype)
    cube4[rng.integers(0, T4, 40), rng.integers(0, ny, 40), rng.integers(0, nx, 40)] = np.nan
    b4 = np.arange(0, T4 + 1, 1200, dtype=np.int64)
    d4 = torch_cuda.from_numpy(cube4).cuda()
    plan4 = hip.FusedPlan(T4, ny * nx, code, b4, np.arange(5), cols)
    assert "packed-counts32" in plan4.describe(), plan4.describe()
    want4 = cport.block_bins(cube4, b4, dda).reshape(4, -1, 13)
    d_ = plan4.run(d4, csr, want_cells=False)
    assert "last-run=count-gather/16-lane" in plan4.describe(), plan4.describe()          # four periods, rows of ~21 entries
    p_ = plan4.run(d4, csr, want_cells=True)
    np.testing.assert_array_equal(np.transpose(p_["cells"].cpu().numpy(), (1, 2, 0)), want4)
    # (four periods, rows of a few dozen entries: the gather gives a (row, period) pair to a group of lanes and adds their shares in a
    # fixed tree — k_csr_spmm_counts_sub, fewer periods than entries per row —, so it equals the table-order sums to rounding; with one
    # lane per pair, AFHIP_COUNTS_SPMM_SUB=0, bit for bit; every group size the same numbers)
    for key in ("num", "den", "res"):
        np.testing.assert_allclose(d_[key].cpu().numpy(), p_[key].cpu().numpy(), rtol=1e-14, equal_nan=True, err_msg=key)
    # ... and with more periods than a row has entries the gather keeps one lane per pair, the table's order (configs[3]: 17 entries, 251 periods)
    b48 = np.arange(0, T4 + 1, 100, dtype=np.int64)
    plan48 = hip.FusedPlan(T4, ny * nx, code, b48, np.arange(49), cols)
    d48, p48 = plan48.run(d4, csr, want_cells=False), None
    assert "last-run=count-gather/1-lane" in plan48.describe(), plan48.describe()
    p48 = plan48.run(d4, csr, want_cells=True)
    assert "count-gather" not in plan48.describe(), plan48.describe()
    for key in ("num", "den", "res"):
        np.testing.assert_array_equal(d48[key].cpu().numpy(), p48[key].cpu().numpy(), err_msg=key)
    for sub in ("0", "4", "8", "16"):
        monkeypatch.setenv("AFHIP_COUNTS_SPMM_SUB", sub)
        ps = hip.FusedPlan(T4, ny * nx, code, b4, np.arange(5), cols)
        monkeypatch.delenv("AFHIP_COUNTS_SPMM_SUB")
        ds = ps.run(d4, csr, want_cells=False)
        for key in ("num", "den", "res"):
            if sub == "0":
                np.testing.assert_array_equal(ds[key].cpu().numpy(), p_[key].cpu().numpy(), err_msg=key)
            else:
                np.testing.assert_allclose(ds[key].cpu().numpy(), p_[key].cpu().numpy(), rtol=1e-14, equal_nan=True, err_msg=key + sub)
    # shuffled slot order and a two-level use (daily mean + annual bins on raw hourly-like groups)
    perm = rng.permutation(13)
    plan2 = hip.FusedPlan(T, ny * nx, code, bounds, np.arange(4), [cols[i] for i in perm] + [dict(inner="mean")])
    got2 = plan2.run_temporal(d).cpu().numpy()
    np.testing.assert_array_equal(got2[:13], got[perm])
    # unequal widths are not a histogram: integer-counter path, same numbers
    odd = [[-20, 0, 0], [0, 7.5, 0], [7.5, 10, 0], [10, 30, 0], [30, 99, 0]]
    plan3 = hip.FusedPlan(T, ny * nx, code, bounds, np.arange(4), [dict(inner="bins", inner_args=r) for r in odd])
    assert "_hist" not in plan3.describe() and "_ibins" in plan3.describe()
    np.testing.assert_array_equal(np.transpose(plan3.run_temporal(d).cpu().numpy(), (1, 2, 0)),
                                  cport.block_bins(cube, bounds, odd).reshape(3, -1, 5))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("part", [(-20.0, 5.0, 13), (0.0, 1.0, 16), (-7.5, 2.5, 12), (250.0, 10.0, 8), (-1.0, 0.25, 9), (-64.0, 8.0, 16),
                                  (1024.0, 0.5, 6), (-3.0, 0.125, 11)])
def test_arithmetic_edge_histogram_one_sided_guess_on_every_edge(torch_cuda, dtype, part):
    """The arithmetic-edge histogram guesses a value's bin with a constant biased DOWN (chosen by the host per partition, checked
    on every edge with the kernel's own fma) and repairs upward only (`ha_update`): a value ON an edge must guess the bin below
    and land in no bin, values 1 .. 3 ulps to either side in the neighbouring bins — for every edge of several exactly
    representable partitions (first edge, width, bins), with out-of-range values of every size, signed zeros, NaN and
    infinities in the guard bins.  Bit-exact against the oracle's compare chain (nb_kernels.py:182-199)."""
    from aggfly_amd import hip
    e0, w, n = part
    edges = (e0 + w * np.arange(n + 1)).astype(np.float64)
    assert np.all(edges.astype(dtype).astype(np.float64) == edges)          # exactly representable: the arithmetic form's premise
    T, ny, nx = 400, 4, 16
    rng = np.random.default_rng(int(abs(e0) * 8 + n))
    cube = rng.uniform(edges[0] - 2 * w, edges[-1] + 2 * w, (T, ny, nx)).astype(dtype)
    flat = cube.reshape(-1)
    plant = []
    for e in edges.astype(dtype):
        up = dn = e
        plant.append(e)
        for _ in range(3):
            up = np.nextafter(up, dtype(np.inf)); dn = np.nextafter(dn, dtype(-np.inf))
            plant += [up, dn]
    plant += [dtype(v) for v in (0.0, -0.0, 1e-30, -1e-30, 1e30, -1e30, np.inf, -np.inf, np.nan,
                                 np.finfo(dtype).max, -np.finfo(dtype).max, np.finfo(dtype).tiny, edges[0] - 0.5 * w, edges[-1] + 0.5 * w)]
    plant = np.array(plant * 6, dtype=dtype)
    flat[rng.choice(flat.size, plant.size, replace=False)] = plant
    bounds = np.array([0, 150, 400], dtype=np.int64)
    dda = [[edges[i], edges[i + 1], 0] for i in range(n)]
    want = cport.block_bins(cube, bounds, dda)                               # [G, ny, nx, D]
    d = torch_cuda.from_numpy(cube).cuda()
    plan = hip.FusedPlan(T, ny * nx, hip.F64 if dtype == np.float64 else hip.F32, bounds, np.arange(3),
                         [dict(inner="bins", inner_args=r) for r in dda], exact_order=True)
    assert "_hist_arith" in plan.describe(), plan.describe()
    got = plan.run_temporal(d).cpu().numpy()                                 # [D, G, cells]
    np.testing.assert_array_equal(np.transpose(got, (1, 2, 0)), want.reshape(2, -1, n))
    # every planted edge value really sits in no bin: the counts of a cell add up to its steps minus its edge / NaN / out-of-range values
    inside = (cube > edges[0]) & (cube < edges[-1]) & ~np.isin(cube, edges.astype(dtype))
    np.testing.assert_array_equal(got[:, 0].sum(axis=0), inside[:150].reshape(150, -1).sum(axis=0))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("edges", ["tenths", "kelvin"])
def test_lds_histogram_with_edges_that_float32_cannot_represent(torch_cuda, dtype, edges):
    """The histogram path guesses the bin in the input precision and repairs the guess against the
    exact edges.  Edges like 0.1*k or 250.15 + 0.7*k have no float32 representation: the value
    next to such an edge on either side, quantised data sitting on every edge, and the guard bins
    below / above the partition must all count exactly as the reference's double compares do."""
    from aggfly_amd import hip
    T, ny, nx = 400, 4, 32
    rng = np.random.default_rng(5)
    if edges == "tenths":
        e = 0.1 * np.arange(-3, 11)                                      # 13 bins of width 0.1 (0.1*k is inexact)
        cube = np.round(rng.normal(0.4, 0.5, (T, ny, nx)), 1)            # data quantised to the edge lattice
    else:
        e = 250.15 + 0.7 * np.arange(0, 17)                              # 16 bins in kelvin
        cube = rng.normal(256, 5, (T, ny, nx))
    cube = cube.astype(dtype)
    flat = cube.reshape(-1)
    pick = rng.choice(flat.size, 900, replace=False)
    ee = rng.choice(e, 900)
    flat[pick[:300]] = ee[:300].astype(dtype)                            # the nearest representable value of an edge
    flat[pick[300:600]] = np.nextafter(ee[300:600].astype(dtype), dtype(np.inf))
    flat[pick[600:880]] = np.nextafter(ee[600:880].astype(dtype), dtype(-np.inf))
    flat[pick[880:890]] = np.nan
    flat[pick[890:895]] = np.inf
    flat[pick[895:900]] = -np.inf
    bounds = np.array([0, 150, T], dtype=np.int64)
    dda = [[e[i], e[i + 1], 0] for i in range(len(e) - 1)]
    want = cport.block_bins(cube, bounds, dda).reshape(2, -1, len(dda))
    d = torch_cuda.from_numpy(cube).cuda()
    code = hip.F64 if dtype == np.float64 else hip.F32
    cols = [dict(inner="bins", inner_args=r) for r in dda]
    for tuning in (0, 208, 104):                                         # default, two cells per lane, shallow prefetch
        plan = hip.FusedPlan(T, ny * nx, code, bounds, np.arange(3), cols, tuning=tuning)
        if tuning == 0 or (tuning == 208 and dtype == np.float32):      # other arms are hints and may pick plain counters
            assert "_hist" in plan.describe(), plan.describe()
        got = plan.run_temporal(d).cpu().numpy()
        np.testing.assert_array_equal(np.transpose(got, (1, 2, 0)), want, err_msg=f"tuning {tuning}")
    assert want.sum() > 0.5 * T * ny * nx * (0.3 if edges == "tenths" else 0.5)      # the bins really are populated


def test_lds_histogram_is_refused_for_bins_narrower_than_the_data_resolution(torch_cuda):
    """Bins much narrower than a float32 ulp of the edges cannot be guessed to within one bin in
    float32: the plan falls back to the per-bin counters and stays exact."""
    from aggfly_amd import hip
    T, ny, nx = 64, 2, 32
    rng = np.random.default_rng(6)
    e = 300.0 + 1e-5 * np.arange(0, 9)
    cube = (300.0 + rng.uniform(-2e-5, 1e-4, (T, ny, nx))).astype(np.float32)
    bounds = np.array([0, T], dtype=np.int64)
    dda = [[e[i], e[i + 1], 0] for i in range(8)]
    plan = hip.FusedPlan(T, ny * nx, hip.F32, bounds, np.arange(2), [dict(inner="bins", inner_args=r) for r in dda])
    assert "_hist" not in plan.describe()
    got = plan.run_temporal(torch_cuda.from_numpy(cube).cuda()).cpu().numpy()
    np.testing.assert_array_equal(np.transpose(got, (1, 2, 0)), cport.block_bins(cube, bounds, dda).reshape(1, -1, 8))
    plan64 = hip.FusedPlan(T, ny * nx, hip.F64, bounds, np.arange(2), [dict(inner="bins", inner_args=r) for r in dda])
    # (at float64 the edges 300 + 1e-5*k are not equally spaced to 1e-9 of a width either: plain counters again)
    c64 = cube.astype(np.float64)
    got = plan64.run_temporal(torch_cuda.from_numpy(c64).cuda()).cpu().numpy()
    np.testing.assert_array_equal(np.transpose(got, (1, 2, 0)), cport.block_bins(c64, bounds, dda).reshape(1, -1, 8))


def test_many_periods_and_grid_limits(torch_cuda):
    """More output periods than a grid dimension holds (daily single-level output of a long
    series: 70,000 periods) through both the cells-only and the panel path."""
    from aggfly_amd import hip
    T, ny, nx = 70000, 2, 4
    rng = np.random.default_rng(5)
    cube = rng.normal(10, 5, (T, ny, nx)).astype(np.float32)
    cube[rng.integers(0, T, 50), 0, 0] = np.nan
    d = torch_cuda.from_numpy(cube).cuda()
    ib = np.arange(T + 1, dtype=np.int64)
    want = cport.block_stat(cube, ib, "max")
    np.testing.assert_array_equal(hip.group_stat(d, ib, "max").cpu().numpy(), want)
    plan = hip.FusedPlan(T, ny * nx, hip.F32, ib, ib, [dict(inner="max"), dict(inner="dd", inner_args=(5, 15, 0))])
    cells = plan.run_temporal(d).cpu().numpy()
    np.testing.assert_array_equal(cells[0].reshape(T, ny, nx), want.astype(np.float64))
    csr = hip.CSR([0, 0, 1, 1, 1], [0, 1, 2, 5, 7], [0.5, 0.5, 0.2, 0.3, 0.5], 2, ny * nx)
    out = plan.run(d, csr)                       # default: 140,000 (row, period) pairs of 8 lanes gather the slots directly
    exact = hip.FusedPlan(T, ny * nx, hip.F32, ib, ib, [dict(inner="max"), dict(inner="dd", inner_args=(5, 15, 0))],
                          exact_order=True).run(d, csr)              # table order: combine + one thread per (row, column)
    flat = want.astype(np.float64).reshape(T, -1)
    dd = cport.block_dd(cube, ib, [5, 15, 0])[..., 0].astype(np.float64).reshape(T, -1)
    valid = ~(np.isnan(flat) | np.isnan(dd))
    for r, (cs, ws) in enumerate((([0, 1], [0.5, 0.5]), ([2, 5, 7], [0.2, 0.3, 0.5]))):
        den = sum(w * valid[:, c] for c, w in zip(cs, ws))
        num = np.zeros(T)
        for c, w in zip(cs, ws):
            num = num + w * np.where(valid[:, c], flat[:, c], 0.0)
        with np.errstate(invalid="ignore", divide="ignore"):
            want_r = np.where(den != 0, num / den, np.nan)
        np.testing.assert_array_equal(exact["res"][0, r].cpu().numpy(), want_r)
        np.testing.assert_allclose(out["res"][0, r].cpu().numpy(), want_r, rtol=1e-14, equal_nan=True)


def test_many_period_plans_take_one_time_chunk_per_period(torch_cuda, monkeypatch):
    """Plans with eight or more output periods on a grid that would otherwise stream as a few long chunks are cut into one time
    chunk per period (no extra slot: whole periods stay together, so every per-cell value is bit-identical to the few-chunk
    layout); fewer periods keep round 2's rule (`afhip_api.hip:build_chunks`, profiles/r03_period_end_stores.txt)."""
    from aggfly_amd import hip
    torch = torch_cuda
    T, ny, nx = 12 * 64, 512, 512                                   # 512 tiles of 256 threads x 2 cells: round 2's rule makes two chunks
    g = torch.Generator(device="cuda").manual_seed(31)
    d = (15 + 12 * torch.randn((T, ny, nx), generator=g, device="cuda", dtype=torch.float32))
    d[5, 3, 7] = float("nan")
    ib = np.arange(0, T + 1, 16, dtype=np.int64)                     # 48 inner groups of 16 steps
    cols = [dict(inner="mean", transform="pow", transform_arg=2.0, outer="sum"), dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]

    def chunks_slots(plan):
        txt = plan.describe()
        return int(txt.split("chunks=")[1].split()[0]), int(txt.split("out_slots=")[1].split()[0])

    ob12 = np.arange(0, 49, 4, dtype=np.int64)                       # 12 periods of 64 steps
    new = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob12, cols)
    assert chunks_slots(new) == (12, 12), new.describe()
    monkeypatch.setenv("AFHIP_NO_PERIOD_CHUNKS", "1")
    old = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob12, cols)
    monkeypatch.delenv("AFHIP_NO_PERIOD_CHUNKS")
    assert chunks_slots(old)[0] < 8 and chunks_slots(old)[1] == 12, old.describe()      # whole periods either way
    got = new.run_temporal(d).cpu().numpy()
    np.testing.assert_array_equal(got, old.run_temporal(d).cpu().numpy())
    ob4 = np.arange(0, 49, 12, dtype=np.int64)                       # four periods: too few chunks to be worth their last round
    few = hip.FusedPlan(T, ny * nx, hip.F32, ib, ob4, cols)
    assert chunks_slots(few)[0] < 8 and chunks_slots(few)[1] == 4, few.describe()
    # the oracle on the first four grid rows (2,048 cells), on the same values as float64: the plan was built without the
    # reference's float32 intermediates (`rounding`), so its arithmetic is float64 from the first add on
    band = d[:, :4, :].cpu().numpy().astype(np.float64)
    want = _oracle_two_level(band, ib, ob12, cols)
    np.testing.assert_array_equal(got[1][:, :4 * nx], want[1])       # dd -> sum: bit-exact
    np.testing.assert_allclose(got[0][:, :4 * nx], want[0], rtol=4e-16, equal_nan=True)      # mean^2 -> sum


def test_caller_workspace_and_side_stream(torch_cuda):
    """The ABI's ownership rules: work is enqueued on the caller's stream and uses a caller-owned workspace (the Python host hands
    every plan a block of torch's caching allocator by default; `describe()` names which it was) or, for a bare C caller, scratch
    the library allocates itself; same results; a workspace that is too small, misaligned or on the host is refused."""
    from aggfly_amd import hip
    torch = torch_cuda
    T, ny, nx = 24 * 50, 10, 16
    cube = torch.from_numpy(_cube(T, ny, nx, np.float64, seed=17)).cuda()
    ib = synth.hourly_bounds(T)
    ob = np.array([0, 20, 50], dtype=np.int64)
    cols = [dict(inner="mean", outer="sum"), dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    tab = synth.weights_table(ny, nx, 6, seed=18)
    csr = hip.CSR(tab.index_right.to_numpy(), tab.cell_id.to_numpy(), tab.weight.to_numpy(), int(tab.index_right.max()) + 1, ny * nx)
    plan = hip.FusedPlan(T, ny * nx, hip.F64, ib, ob, cols)
    ref = plan.run(cube, csr)["res"].clone()
    assert "(caller-owned)" in plan.describe()
    own = plan.run(cube, csr, workspace="library")["res"].clone()
    assert "(plan-owned hipMalloc)" in plan.describe()
    assert torch.equal(torch.nan_to_num(own, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))
    assert plan.workspace_bytes(csr) > plan.workspace_bytes() > 0
    ws = torch.empty(plan.workspace_bytes(csr), dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = plan.run(cube, csr, workspace=ws)["res"]
    side.synchronize()
    assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))
    with pytest.raises(ValueError, match="workspace"):
        plan.run(cube, csr, workspace=torch.empty(16, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError, match="workspace"):                 # sized for the temporal stage only: the library checks it too
        plan.run(cube, csr, workspace=torch.empty(plan.workspace_bytes(), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError, match="256-byte aligned"):
        plan.run(cube, csr, workspace=torch.empty(plan.workspace_bytes(csr) + 512, dtype=torch.uint8, device="cuda")[8:])
    # a second table with more rows on the same plan: the plan-owned scratch grows without a free on the run path, the torch block is re-made
    tab2 = synth.weights_table(ny, nx, 40, seed=19)
    csr2 = hip.CSR(tab2.index_right.to_numpy(), tab2.cell_id.to_numpy(), tab2.weight.to_numpy(), int(tab2.index_right.max()) + 1, ny * nx)
    a = plan.run(cube, csr2)["res"].clone()
    b = plan.run(cube, csr2, workspace="library")["res"]
    assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))
    assert torch.equal(torch.nan_to_num(plan.run(cube, csr)["res"], nan=-1.0), torch.nan_to_num(ref, nan=-1.0))


def test_c_abi_from_plain_c(torch_cuda, tmp_path):
    """The boundary is a C ABI: a C11 program compiled with gcc links libaggfly_hip.so, runs a
    fused plan + CSR reduce and checks it against its own host loop (examples/c_abi_demo.c)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_demo")
    lib = os.path.join(root, "aggfly_amd")
    subprocess.check_call(["gcc", "-std=c11", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "c_abi_demo.c"),
                           "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", "-L" + lib, "-laggfly_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C ABI OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("dtype,odtype", [(np.float64, np.float64), (np.float32, np.float32), (np.float32, np.float64)])
def test_fused_inter_column_bit_exact(torch_cuda, dtype, odtype):
    """AFHIP_TF_INTER (`Dataset.interact`, dataset.py:483-518,547-563) at the group end of the streaming kernel: mean@date
    times a second cube [G1, cells], then sum@month — against the oracle's np.multiply between the two levels, bit for bit
    (exact_order), on ragged row lengths (scalar fallback) and on rows that take the vector arms; an unbound column fails."""
    from aggfly_amd import hip
    for (T, ny, nx) in ((24 * 70 + 5, 5, 7), (24 * 70 + 5, 6, 20)):
        cube = _cube(T, ny, nx, dtype, seed=21)
        ib = np.concatenate([np.arange(0, T, 24), [T]]).astype(np.int64)
        G1 = len(ib) - 1
        ob = np.unique(np.concatenate([np.arange(0, G1, 30), [G1]])).astype(np.int64)
        other = np.random.default_rng(22).normal(1.0, 0.6, (G1, ny, nx)).astype(odtype)
        other[2, 1, 3] = np.nan
        cols = [dict(inner="mean", transform="inter", outer="sum"), dict(inner="max", transform="inter", outer="max"),
                dict(inner="mean", outer="sum")]
        plan = hip.FusedPlan(T, ny * nx, hip.F64 if dtype == np.float64 else hip.F32, ib, ob, cols, exact_order=True)
        d = torch_cuda.from_numpy(cube).cuda()
        with pytest.raises(ValueError, match="never bound"):
            plan.run_temporal(d)
        od = torch_cuda.from_numpy(other).cuda()
        plan.bind_inter(0, od)
        plan.bind_inter(1, od)
        with pytest.raises(ValueError, match="no inter transform"):
            plan.bind_inter(2, od)
        got = plan.run_temporal(d).cpu().numpy().reshape(3, len(ob) - 1, ny, nx)
        m = cport.resample(cube.astype(np.float64), ib, "mean")
        mx = cport.resample(cube.astype(np.float64), ib, "max")
        o64 = other.astype(np.float64)
        np.testing.assert_array_equal(got[0], cport.resample(np.multiply(m, o64), ob, "sum"))
        np.testing.assert_array_equal(got[1], cport.resample(np.multiply(mx, o64), ob, "max"))
        np.testing.assert_array_equal(got[2], cport.resample(m, ob, "sum"))


def test_skewed_rows_take_the_wave_and_segment_paths(torch_cuda):
    """Real admin-2 tables span four decades of row lengths.  On a log-normal table (rows of 2 .. 50,000 entries) the
    default spatial stage runs one wave per row segment (few output columns) or one thread per (segment, column) and adds
    the pieces of cut rows in order — a different association than np.add.at's table order, so 1e-12 against the oracle's
    `_scatter_block` restatement (spatial.py:181-199); `afhip_scatter_block` and exact_order plans keep the table order
    bit for bit."""
    from aggfly_amd import hip
    ny, nx, R = 300, 500, 800
    tab = synth.weights_table(ny, nx, R, seed=11, skew="lognormal", zero_frac=0.01)
    lens = tab.groupby("index_right").size()
    assert lens.max() >= 40_000 and lens.min() <= 3 and (lens > 1024).sum() >= 5             # cut rows exist
    ridx, cidx, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
    nR = int(ridx.max()) + 1
    csr = hip.CSR(ridx, cidx, w, nR, ny * nx)
    rng = np.random.default_rng(12)
    for K, nt in ((1, 1), (2, 1), (5, 1), (3, 3), (2, 10)):          # Q = 2, 3, 6, 12 (one wave per segment) and 30 (thread per column)
        x = rng.normal(20, 5, (K, ny * nx, nt))
        x[0, rng.integers(0, ny * nx, 500), rng.integers(0, nt, 500)] = np.nan
        nums, den, ids = spatial_num_den({f"k{k}": x[k] for k in range(K)}, tab, np.arange(ny * nx))
        gn, gd, gr = csr.wavg(torch_cuda.from_numpy(x).cuda())
        np.testing.assert_allclose(gd.cpu().numpy(), den, rtol=1e-12, atol=0)
        for k in range(K):
            np.testing.assert_allclose(gn[k].cpu().numpy(), nums[f"k{k}"], rtol=1e-12, atol=1e-9)
        with np.errstate(invalid="ignore", divide="ignore"):
            want = np.where(den != 0, np.stack([nums[f"k{k}"] for k in range(K)]) / den, np.nan)
        np.testing.assert_allclose(gr.cpu().numpy(), want, rtol=1e-11, equal_nan=True)
    block = rng.normal(20, 5, (ny * nx, 3))
    np.testing.assert_array_equal(csr.scatter_block(torch_cuda.from_numpy(block).cuda()).cpu().numpy(), ref_scatter(block, ridx, cidx, w, nR))
    # a fused plan: exact_order = table order bit for bit; the default = the same numbers to rounding
    T = 24 * 20
    cube = _cube(T, ny, nx, np.float64, seed=13)
    ib = synth.hourly_bounds(T)
    ob = np.array([0, len(ib) - 1], dtype=np.int64)
    cols = [dict(inner="mean", outer="sum"), dict(inner="dd", inner_args=(10, 30, 0), outer="sum")]
    d = torch_cuda.from_numpy(cube).cuda()
    ex = hip.FusedPlan(T, ny * nx, hip.F64, ib, ob, cols, exact_order=True).run(d, csr, want_cells=True)
    cells = ex["cells"].cpu().numpy()                                 # [K, 1, C]
    valid = ~np.isnan(cells).any(axis=0)[0]
    for k in range(2):
        np.testing.assert_array_equal(ex["num"][k, :, 0].cpu().numpy(),
                                      ref_scatter(np.where(valid, cells[k, 0], 0.0)[:, None], ridx, cidx, w, nR)[:, 0])
    fast = hip.FusedPlan(T, ny * nx, hip.F64, ib, ob, cols).run(d, csr)
    np.testing.assert_allclose(fast["res"].cpu().numpy(), ex["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)
    # bin counts (packed records gathered directly, segments of cut rows added afterwards) == the panel route
    Td = 365 * 3
    daily = _cube(Td, ny, nx, np.float32, seed=14)[::1]
    edges = np.arange(-20, 50, 5.0)
    bcols = [dict(inner="bins", inner_args=(edges[i], edges[i + 1], 0)) for i in range(13)]
    yb = np.array([0, 365, 730, 1095], dtype=np.int64)
    plan = hip.FusedPlan(Td, ny * nx, hip.F32, yb, np.arange(4, dtype=np.int64), bcols)
    assert "packed-counts" in plan.describe()
    dd = torch_cuda.from_numpy(daily).cuda()
    direct = plan.run(dd, csr)["res"].cpu().numpy()
    via_panel = plan.run(dd, csr, want_cells=True)["res"].cpu().numpy()
    np.testing.assert_allclose(direct, via_panel, rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("table", ["county", "tiny", "skewed"])
def test_slot_gather_spatial_stage_against_the_panel_route_and_the_oracle(torch_cuda, monkeypatch, table):
    """The default spatial stage of a plan run without per-cell output gathers the period slots directly (k_csr_spmm_slots:
    slot merge + shared validity + weighted sums in one pass, no cell-major panel).  Checked against (a) the route it replaced
    (k_combine_slots + k_csr_spmm*, AFHIP_NO_SLOT_SPMM=1): bit for bit where a whole wave serves a (segment, period) and the old
    route was the wave kernel, 1e-12 otherwise; (b) exact_order plans (table order, spatial.py:181-186) at 1e-12; (c) the
    oracle's spatial restatement on the plan's own per-cell values.  Tables: county-sized rows, rows of a handful of cells
    (8 lanes per pair), log-normal rows with cut segments.  Plans: several periods incl. an empty one, periods cut into
    several slots, outer mean / min / max, float32 rounding of the final value, NaN cells (shared validity), K = 1 .. 13."""
    from aggfly_amd import hip
    if table == "county":
        ny, nx, R = 96, 160, 60
        tab = synth.weights_table(ny, nx, R, seed=21, zero_frac=0.05)
    elif table == "tiny":
        ny, nx, R = 40, 64, 600
        tab = synth.weights_table(ny, nx, R, seed=22, secondary=True)
    else:
        ny, nx, R = 200, 300, 300
        tab = synth.weights_table(ny, nx, R, seed=23, skew="lognormal")
    C = ny * nx
    ridx, cidx, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
    nR = int(ridx.max()) + 1
    csr = hip.CSR(ridx, cidx, w, nR, C)
    mean_len = len(w) / nR
    assert {"county": mean_len > 64, "tiny": mean_len <= 8, "skewed": True}[table]
    T = 24 * 60
    cube = _cube(T, ny, nx, np.float64, seed=24)
    d = torch_cuda.from_numpy(cube).cuda()
    ib = synth.hourly_bounds(T)
    poly = [dict(inner="mean", transform="pow", transform_arg=float(e), outer="sum") for e in (1, 2, 3, 4)]
    dd = dict(inner="dd", inner_args=(10, 30, 0), outer="sum")
    plans = [
        # (outer bounds over the 60 daily groups, columns)
        (np.array([0, 60]), [dict(inner="mean", outer="sum")]),                                   # P = 1: the headline's shape
        (np.array([0, 60]), [dd] + poly),
        (np.array([0, 10, 10, 31, 60]), poly),                                                     # an empty period
        (np.array([0, 7, 20, 33, 47, 60]), [dict(inner="mean", outer="mean"), dict(inner="max", outer="max"),
                                            dict(inner="min", outer="min"), dd,
                                            dict(inner="mean", outer="mean", rounding=hip.ROUND_FINAL)]),
        (np.arange(0, 61, 5), [dict(inner="dd", inner_args=(float(t), float(t) + 7, 0), outer="sum") for t in range(-10, 29, 3)]),   # K = 13
    ]
    saw_split = False
    for ob, cols in plans:
        ob = ob.astype(np.int64)
        K, P = len(cols), len(ob) - 1
        plan = hip.FusedPlan(T, C, hip.F64, ib, ob, cols)
        slots = int(plan.describe().split("out_slots=")[1].split()[0])
        saw_split = saw_split or slots > P
        fused = plan.run(d, csr)
        monkeypatch.setenv("AFHIP_NO_SLOT_SPMM", "1")            # (experiment knobs are read when a plan is created)
        panel = hip.FusedPlan(T, C, hip.F64, ib, ob, cols).run(d, csr)
        monkeypatch.delenv("AFHIP_NO_SLOT_SPMM")
        whole_wave = mean_len > 32 and (K + 1) * P <= 16            # 64 lanes per pair, and the old route was k_csr_spmm_wave
        for key in ("num", "den", "res"):
            a, b = fused[key].cpu().numpy(), panel[key].cpu().numpy()
            if whole_wave:
                np.testing.assert_array_equal(a, b)
            else:
                np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-9 if key == "num" else 0, equal_nan=True)
        ex = hip.FusedPlan(T, C, hip.F64, ib, ob, cols, exact_order=True).run(d, csr, want_cells=True)
        np.testing.assert_allclose(fused["res"].cpu().numpy(), ex["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(fused["den"].cpu().numpy(), ex["den"].cpu().numpy(), rtol=1e-12)
        # the oracle's spatial stage on the default plan's own per-cell values (these may differ from the exact_order
        # plan's in the last bits where a period is cut into chunks)
        cells = plan.run(d, csr, want_cells=True)["cells"].cpu().numpy()                        # [K, P, C]
        nums, den, _ = spatial_num_den({f"k{k:02d}": cells[k].T for k in range(K)}, tab, np.arange(C))
        np.testing.assert_allclose(fused["den"].cpu().numpy(), den, rtol=1e-12)
        for k in range(K):
            np.testing.assert_allclose(fused["num"][k].cpu().numpy(), nums[f"k{k:02d}"], rtol=1e-12, atol=1e-9)
    assert saw_split, "no plan of this test cut a period into several slots: the merge path went untested"
    # every group width gives the same sums to rounding
    ob = np.array([0, 20, 41, 60], dtype=np.int64)
    plan = hip.FusedPlan(T, C, hip.F64, ib, ob, [dd] + poly)
    want = plan.run(d, csr)["num"].cpu().numpy()
    for sub in (8, 16, 32, 64):
        for order in ("v", "p"):                                  # ... and either order of the (segment, period) pairs over the grid, bit for bit
            monkeypatch.setenv("AFHIP_SLOT_SPMM_SUB", str(sub))
            monkeypatch.setenv("AFHIP_SLOT_SPMM_ORDER", order)
            got = hip.FusedPlan(T, C, hip.F64, ib, ob, [dd] + poly).run(d, csr)["num"].cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9)
            if order == "v":
                first = got
            else:
                np.testing.assert_array_equal(got, first)
    monkeypatch.delenv("AFHIP_SLOT_SPMM_SUB")
    monkeypatch.delenv("AFHIP_SLOT_SPMM_ORDER")


@pytest.mark.parametrize("kind", ["hourly_f64", "hourly_f32", "daily_f32", "hourly_f64_lognormal", "six_hourly_f32", "six_hourly_f64", "pairs_poly_f32",
                                  "pairs_sine_f32", "pairs_two_f64", "dd_only_f32", "daily_multi_dd_f32", "eight_hourly_f32", "gaps_f32"])
def test_region_fused_period_ends_against_the_per_cell_routes(torch_cuda, monkeypatch, kind):
    """Plans with several output periods, sum-like outer reducers and no per-cell output reduce their cells by region INSIDE the
    streaming kernel at every period end (FusedArgs::rf_w: per-run weighted sums from a wave-private LDS block, k_rf_reduce adds a
    region's runs) — the per-cell period values are never written.  Same numbers as the routes that do write them
    (AFHIP_NO_REGION_FUSED=1: k_csr_spmm_slots; exact_order: table order) to rounding, and as the oracle's spatial stage on the
    plan's own per-cell values; NaN cells (shared validity), zero-weight regions, an empty period, border cells that sit in two
    regions, cells in up to five regions (the "extras"), regions of a few cells, the last partly filled tile.  Plans the route does not
    cover (float32 rounding of the final value, one period, more than six columns or four threshold slots) stay on the per-cell routes.
    Round 4's twins (the last five kinds): the sine-only and two-column pair forms — taken when the per-cell period values would be
    5 % of the cube or more —, threshold-only plans, a single-level panel of three degree-day columns, 8-hourly data (three-row groups)."""
    from aggfly_amd import hip
    ny, nx, R = 71, 130, 40                                          # 9,230 cells: a last tile that is partly filled
    if kind.endswith("lognormal"):
        # region sizes over three decades (one region of a third of the grid, spanning dozens of wave tiles; regions of one cell)
        tab = synth.weights_table(ny, nx, 60, seed=45, skew="lognormal", zero_frac=0.05)
        sizes = tab.groupby("index_right").size()
        assert sizes.max() > 2000 and sizes.min() <= 4
        kind = "hourly_f64"
    else:
        tab = synth.weights_table(ny, nx, R, seed=41, zero_frac=0.05)
    C = ny * nx
    ridx, cidx, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
    nR = int(ridx.max()) + 1
    assert tab.groupby("cell_id").size().max() == 2                  # border cells belong to two regions
    csr = hip.CSR(ridx, cidx, w, nR, C)
    poly = [dict(inner="mean", transform="pow", transform_arg=float(e), outer="sum") for e in (1, 2, 3, 4)]
    if kind.startswith("hourly"):
        dtype = np.float64 if kind.endswith("f64") else np.float32
        T, spd = 24 * 60, 24
        cube = _cube(T, ny, nx, dtype, seed=42)
        # float64: the configs[1] columns (a threshold slot + the polynomial); float32 plans with threshold slots stay on the
        # per-cell route (measured behind), so the float32 case is the polynomial alone — the reference's own benchmark plan
        cols = ([dict(inner="dd", inner_args=(10, 30, 0), outer="sum")] if dtype == np.float64 else []) + poly
    elif kind.startswith("six_hourly") or kind.startswith("pairs"):
        # short inner groups: the lean four-row form (every one of them has a twin) and the six-column lean pair form (plans of three
        # columns and more; the sine-only and one- / two-column pair forms stay on the per-cell route)
        dtype = np.float64 if kind.endswith("f64") else np.float32
        spd = 4 if kind.startswith("six") else 2
        T = spd * 60
        cube = _cube(T, ny, nx, dtype, seed=47)
        cols = poly if spd == 2 else [dict(inner="mean", transform="pow", transform_arg=float(e), outer="sum") for e in (1, 3)] + [dict(inner="max", outer="sum")]
        if kind == "pairs_sine_f32":
            cols = [dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum")]
        if kind == "pairs_two_f64":
            cols = [dict(inner="mean", outer="sum"), dict(inner="max", outer="mean")]
    elif kind == "eight_hourly_f32":
        dtype, spd = np.float32, 3                                   # three-row groups: the lean `_tri` form
        T = spd * 60
        cube = _cube(T, ny, nx, dtype, seed=49)
        cols = [dict(inner="mean", transform="pow", transform_arg=float(e), outer="sum") for e in (1, 2)] + [dict(inner="min", outer="mean")]
    elif kind == "gaps_f32":
        dtype, spd = np.float32, 4                                   # 6-hourly data with missing steps: groups of two to four rows, the `_rag` form
        T = spd * 60 - 17
        cube = _cube(T, ny, nx, dtype, seed=50)
        cols = [dict(inner="mean", transform="pow", transform_arg=float(e), outer="sum") for e in (1, 2)] + [dict(inner="max", outer="mean")]
    elif kind in ("dd_only_f32", "daily_multi_dd_f32"):
        dtype = np.float32
        T, spd = 24 * 60, 24
        cube = _cube(T, ny, nx, dtype, seed=48)
        cols = {"dd_only_f32": [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")],                      # no statistic at all: one threshold slot
                "daily_multi_dd_f32": [dict(inner="dd", inner_args=(float(t), float(t) + 9, t % 2)) for t in (0, 10, 20)]}[kind]
    else:
        dtype, spd = np.float32, 8                                   # three-hourly steps: min / max sources beside the mean
        T = spd * 60
        cube = _cube(T, ny, nx, dtype, seed=43)
        cols = [dict(inner="max", outer="max"), dict(inner="min", transform="pow", transform_arg=2.0, outer="sum"), dict(inner="mean", outer="mean"),
                dict(inner="mean", outer="dd", outer_args=(10, 30, 0)), dict(inner="min", outer="min")]
    code = hip.F64 if dtype == np.float64 else hip.F32
    d = torch_cuda.from_numpy(cube).cuda()
    ib = synth.hourly_bounds(T, spd)                                 # 60 inner groups
    if kind == "gaps_f32":
        lens = np.full(60, 4)
        lens[np.random.default_rng(51).choice(60, 13, replace=False)] = 3
        lens[[7, 31]] = 2
        ib = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        assert ib[-1] == T
    # periods short enough to stay whole on this small grid (a period cut into several slots rules the route out): twelve or
    # six periods, the second one empty
    ob = (np.array([0, 5, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 55, 60], dtype=np.int64) if spd == 24
          else np.array([0, 9, 9, 20, 33, 47, 60], dtype=np.int64))
    if kind == "dd_only_f32":
        ob = np.array([0, 1, 1] + list(range(2, 61)), dtype=np.int64)    # a daily panel, the second period empty
    if kind == "daily_multi_dd_f32":
        ob = np.arange(61, dtype=np.int64)                               # single level: one value per day and column
    K = len(cols)
    plan = hip.FusedPlan(T, C, code, ib, ob, cols)
    assert "region-fused-capable" in plan.describe(), plan.describe()
    if kind.startswith("six_hourly"):
        assert "_quad" in plan.describe(), plan.describe()
    if kind == "pairs_poly_f32":
        assert "_pair_lean" in plan.describe(), plan.describe()
    if kind == "pairs_sine_f32":
        assert "_pair_ss" in plan.describe(), plan.describe()
        # ... only where the period values weigh in: two periods of this year of pairs stay on the per-cell route (monthly: 4 % behind)
        two = hip.FusedPlan(T, C, code, ib, np.array([0, 30, 60], dtype=np.int64), cols)
        assert "_pair_ss" in two.describe() and "region-fused" not in two.describe(), two.describe()
    if kind == "pairs_two_f64":
        assert "_k2_" in plan.describe() and "_pair" in plan.describe(), plan.describe()
    if kind == "dd_only_f32":
        assert "_s0_t1_" in plan.describe(), plan.describe()
        # thirteen degree-day columns: no twin (bound by their arithmetic; measured level with the per-cell route)
        k13 = hip.FusedPlan(T, C, code, ib, ob, [dict(inner="dd", inner_args=(float(t), float(t) + 7, 0), outer="sum") for t in range(-10, 29, 3)])
        assert "_k16_" in k13.describe() and "region-fused" not in k13.describe(), k13.describe()
    if kind == "eight_hourly_f32":
        assert "_tri" in plan.describe(), plan.describe()
    if kind == "gaps_f32":
        assert "_rag" in plan.describe(), plan.describe()
    if kind == "daily_multi_dd_f32":
        assert "_sl" not in plan.describe().split()[0], plan.describe()   # the two-level variant (it has a twin), not the single-level one
    fused = plan.run(d, csr)
    assert "last-run=region-fused" in plan.describe(), plan.describe()
    monkeypatch.setenv("AFHIP_NO_REGION_FUSED", "1")
    plain = hip.FusedPlan(T, C, code, ib, ob, cols)
    monkeypatch.delenv("AFHIP_NO_REGION_FUSED")
    assert "region-fused" not in plain.describe(), plain.describe()
    slots = plain.run(d, csr)
    for key in ("num", "den", "res"):
        np.testing.assert_allclose(fused[key].cpu().numpy(), slots[key].cpu().numpy(), rtol=1e-12, atol=1e-9 if key == "num" else 0,
                                   equal_nan=True, err_msg=key)
    if kind != "daily_multi_dd_f32":
        assert np.isnan(fused["res"].cpu().numpy()[:, :, 1]).all() and (fused["den"].cpu().numpy()[:, 1] == 0).all()      # the empty period
    # per-cell values of the same plan (asking for them takes the per-cell route) through the oracle's spatial stage
    with_cells = plan.run(d, csr, want_cells=True)
    assert "last-run=region-fused" not in plan.describe()
    cells = with_cells["cells"].cpu().numpy()                        # [K, P, C]
    nums, den, _ = spatial_num_den({f"k{k}": cells[k].T for k in range(K)}, tab, np.arange(C))
    np.testing.assert_allclose(fused["den"].cpu().numpy(), den, rtol=1e-12)
    for k in range(K):
        np.testing.assert_allclose(fused["num"][k].cpu().numpy(), nums[f"k{k}"], rtol=1e-12, atol=1e-9)
    if kind == "hourly_f32":
        # float32 with a threshold slot (round 3: only from 24 periods on; with the scan form of the period end, like every other form) — 30 periods of two days
        tcols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum")] + poly
        ob30 = np.arange(0, 61, 2, dtype=np.int64)
        p30 = hip.FusedPlan(T, C, code, ib, ob30, tcols)
        f30 = p30.run(d, csr)
        assert "last-run=region-fused" in p30.describe(), p30.describe()
        x30 = hip.FusedPlan(T, C, code, ib, ob30, tcols, exact_order=True).run(d, csr)
        for key in ("num", "den", "res"):
            np.testing.assert_allclose(f30[key].cpu().numpy(), x30[key].cpu().numpy(), rtol=1e-12, atol=1e-9 if key == "num" else 0,
                                       equal_nan=True, err_msg=key)
    if kind == "hourly_f64":
        ex = hip.FusedPlan(T, C, code, ib, ob, cols, exact_order=True).run(d, csr)
        np.testing.assert_allclose(fused["res"].cpu().numpy(), ex["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)
        # what the route does not cover keeps the per-cell routes: a rounded final value, a single period
        for bad_cols, bad_ob, bad_code in (([dict(inner="mean", outer="sum", rounding=hip.ROUND_FINAL)] + poly, ob, code),
                                           (cols, np.array([0, 60], dtype=np.int64), code)):
            pl = hip.FusedPlan(T, C, bad_code, ib, bad_ob, bad_cols)
            assert "region-fused" not in pl.describe(), pl.describe()
        # a single-level plan on the generic variants — a daily panel of daily statistics, 60 periods of one group — takes the route too
        daily = [dict(inner="mean"), dict(inner="max"), dict(inner="dd", inner_args=(10, 30, 0))]
        dp = hip.FusedPlan(T, C, code, ib, np.arange(61, dtype=np.int64), daily)
        fd = dp.run(d, csr)
        assert "last-run=region-fused" in dp.describe(), dp.describe()
        xd = hip.FusedPlan(T, C, code, ib, np.arange(61, dtype=np.int64), daily, exact_order=True).run(d, csr)
        np.testing.assert_allclose(fd["res"].cpu().numpy(), xd["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(fd["den"].cpu().numpy(), xd["den"].cpu().numpy(), rtol=1e-12)
        # cells that sit in three, four and five regions (junctions of polygons; the run tables hold two entries per cell, the rest
        # are "extras" that k_rf_reduce adds from the values those cells write on the side): 300 random cells get up to three more
        # regions each, one of them a NaN ("ocean") cell and one in a zero-weight region
        rj = np.random.default_rng(46)
        jc = rj.choice(C, 300, replace=False)
        jc[0] = int(np.flatnonzero(np.isnan(cube).all(axis=0).reshape(-1))[0])
        more_c = np.repeat(jc, rj.integers(1, 4, 300))
        more_r = rj.integers(0, nR, len(more_c))
        r3 = np.concatenate([ridx, more_r]); c3 = np.concatenate([cidx, more_c]); w3 = np.concatenate([w, rj.uniform(0.05, 0.9, len(more_c))])
        keep = ~pd.DataFrame({"r": r3, "c": c3}).duplicated().to_numpy()        # a (region, cell) pair appears once in a weights table
        r3, c3, w3 = r3[keep], c3[keep], w3[keep]
        o3 = np.argsort(r3, kind="stable")
        assert np.bincount(c3).max() >= 4
        jcsr = hip.CSR(r3[o3], c3[o3], w3[o3], nR, C)
        a = plan.run(d, jcsr)
        assert "last-run=region-fused" in plan.describe(), plan.describe()
        b = plain.run(d, jcsr)
        for key in ("num", "den", "res"):
            np.testing.assert_allclose(a[key].cpu().numpy(), b[key].cpu().numpy(), rtol=1e-12, atol=1e-9 if key == "num" else 0, equal_nan=True, err_msg=key)
        jtab = pd.DataFrame({"index_right": r3[o3], "cell_id": c3[o3], "weight": w3[o3]})
        nums, den, _ = spatial_num_den({f"k{k}": cells[k].T for k in range(K)}, jtab, np.arange(C))
        np.testing.assert_allclose(a["den"].cpu().numpy(), den, rtol=1e-12)
        for k in range(K):
            np.testing.assert_allclose(a["num"][k].cpu().numpy(), nums[f"k{k}"], rtol=1e-12, atol=1e-9)
        # regions of a handful of cells (runs of two or three cells): still the route — it measured ahead down to five-cell regions
        tiny = synth.weights_table(ny, nx, 2500, seed=44)
        tcsr = hip.CSR(tiny["index_right"].to_numpy(), tiny["cell_id"].to_numpy(), tiny["weight"].to_numpy(), int(tiny["index_right"].max()) + 1, C)
        a = plan.run(d, tcsr)
        assert "last-run=region-fused" in plan.describe(), plan.describe()
        b = plain.run(d, tcsr)
        np.testing.assert_allclose(a["res"].cpu().numpy(), b["res"].cpu().numpy(), rtol=1e-12, equal_nan=True)


def test_region_fused_sums_do_not_depend_on_the_launch_shape(torch_cuda, monkeypatch):
    """The region-fused period ends add a run's cells in cell order and a region's runs in run order; runs are cut per wave TILE (64
    cells per lane-cell), not per workgroup or chunk — so single-wave (the default) or four-wave workgroups, one time chunk per period or two
    periods per chunk, a caller-owned workspace or the plan's own, run sums laid slot-major or run-major: the same bits."""
    from aggfly_amd import hip
    torch = torch_cuda
    T, ny, nx = 24 * 24, 160, 512                                    # 1280 single-wave tiles
    g = torch.Generator(device="cuda").manual_seed(61)
    d = 15 + 12 * torch.randn((T, ny, nx), generator=g, device="cuda", dtype=torch.float64)
    d[7, 5, 9] = float("nan")
    tab = synth.weights_table(ny, nx, 50, seed=62, zero_frac=0.04)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, ny * nx)
    ib = synth.hourly_bounds(T)
    ob = np.arange(0, 25, 3, dtype=np.int64)                         # eight periods of three days
    cols = [dict(inner="dd", inner_args=(10, 30, 0), outer="sum"), dict(inner="mean", transform="pow", transform_arg=3.0, outer="sum"),
            dict(inner="max", outer="mean")]

    def run(env, workspace=False):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = hip.FusedPlan(T, ny * nx, hip.F64, ib, ob, cols)
        for k in env:
            monkeypatch.delenv(k)
        ws = torch.empty(plan.workspace_bytes(csr), dtype=torch.uint8, device="cuda") if workspace else "library"
        out = plan.run(d, csr, workspace=ws)
        assert "last-run=region-fused" in plan.describe(), plan.describe()
        return plan.describe(), {k: out[k].cpu().numpy() for k in ("num", "den", "res")}

    base_desc, base = run({})
    assert "wg=64" in base_desc and "chunks=8 " in base_desc, base_desc
    # ... nor on where the run sums are laid (slot-major / run-major: k_rf_reduce adds the same values in the same order)
    for env, ws, must in (({"AFHIP_FORCE_WG": "256"}, False, "wg=256"), ({"AFHIP_NO_PERIOD_CHUNKS": "1", "AFHIP_NO_ROUND_FILL": "1"}, False, "chunks=4 "), ({}, True, "wg=64"),
                          ({"AFHIP_RF_LAYOUT": "run"}, False, "wg=64"), ({"AFHIP_RF_LAYOUT": "slot"}, True, "wg=64")):
        desc, got = run(env, ws)
        assert must in desc, desc
        for k in base:
            np.testing.assert_array_equal(got[k], base[k], err_msg=f"{env} {k}")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(2 * 150, 6, 20), (2 * 150, 5, 7)])
def test_pair_plans_lean_and_generic_group_ends(torch_cuda, dtype, shape):
    """(tmin, tmax) pairs: every inner group is two rows (configs[4]; the reference's daily mean of tmin and tmax,
    `temporal.py:99-125`).  Plans whose columns are  mean | sum | min | max | sine_dd -> (integer power) -> sum | mean  take the
    lean group end (FEAT bit 8); a column that needs anything else (hinge, an outer min, float32 rounding) sends the plan to the
    generic pair form or, for sum-only plans, off the pair path.  All of them against the oracle: statistics bit-exact (same
    operation order: s = u + v, mean = s / 2), integer powers within 4e-16 (np.power's libm vs the double-double chain), sine 1e-10;
    NaN pairs give NaN periods; several outer periods in one chunk and periods split over chunks."""
    from aggfly_amd import hip
    T, ny, nx = shape
    rng = np.random.default_rng(17)
    day = 15 + 10 * np.sin(2 * np.pi * np.arange(T // 2) / 365.0)
    cube = np.empty((T, ny, nx))
    cube[0::2] = day[:, None, None] - 5 + rng.normal(0, 3, (T // 2, ny, nx))
    cube[1::2] = day[:, None, None] + 6 + rng.normal(0, 3, (T // 2, ny, nx))
    cube[rng.integers(0, T, 9), rng.integers(0, ny, 9), rng.integers(0, nx, 9)] = np.nan
    cube[:, 1, 2] = np.nan
    cube = cube.astype(dtype)
    ib = np.arange(0, T + 1, 2, dtype=np.int64)
    ob = np.array([0, 20, 21, 150], dtype=np.int64)            # three periods: 20 days, one day, 129 days (long enough to be cut in two)
    d = torch_cuda.from_numpy(cube).cuda()
    f64 = cube.astype(np.float64)

    def want(col):
        x = cport.resample(f64, ib, col["inner"], col.get("inner_args"))
        if col.get("transform") == "pow":
            x = cport.power(x, col["transform_arg"])
        elif col.get("transform") == "hinge":
            x = (x > col["transform_arg"]) * (x - col["transform_arg"])
        return cport.resample(x, ob, col["outer"])

    lean = [dict(inner="mean", outer="sum"), dict(inner="mean", transform="pow", transform_arg=2, outer="sum"),
            dict(inner="mean", transform="pow", transform_arg=3, outer="mean"), dict(inner="sum", outer="sum")]
    lean2 = [dict(inner="min", outer="sum"), dict(inner="max", transform="pow", transform_arg=2, outer="mean"),
             dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum"), dict(inner="mean", outer="mean")]
    generic = [dict(inner="mean", transform="hinge", transform_arg=20.0, outer="sum"), dict(inner="max", outer="max"),
               dict(inner="sine_dd", inner_args=(0, 18, 1), outer="sum")]
    sum_only_generic = [dict(inner="mean", outer="max"), dict(inner="sum", transform="hinge", transform_arg=20.0, outer="sum")]
    sine_only = [dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum"), dict(inner="sine_dd", inner_args=(0, 18, 1), outer="mean")]
    light = [dict(inner="mean", outer="sum")]                  # one or two mean / sum columns stream through the LDS-DMA ring instead
    # exponents below 1 are kept off the lean form (its written-out power chain starts at the square; no division code in it)
    odd_powers = [dict(inner="max", transform="pow", transform_arg=-1, outer="sum"), dict(inner="min", transform="pow", transform_arg=0, outer="sum"),
                  dict(inner="mean", transform="pow", transform_arg=5, outer="sum"), dict(inner="mean", transform="pow", transform_arg=1, outer="sum")]
    high_powers = [dict(inner="mean", transform="pow", transform_arg=e, outer="mean") for e in (5, 7, 8)] + [dict(inner="max", outer="sum")]
    for cols, expect in ((lean, "_pair_lean"), (lean2, "_pair_lean"), (sine_only, "_pair_ss"), (generic, "_pair"), (sum_only_generic, None), (light, None),
                         (odd_powers, "_pair"), (high_powers, "_pair_lean")):
        plan = hip.FusedPlan(T, ny * nx, hip.F64 if dtype == np.float64 else hip.F32, ib, ob, cols, exact_order=True)
        name = plan.describe().split()[0]
        if expect is None:
            assert "_pair" not in name, name
        else:
            assert name.endswith(expect), name
        got = plan.run_temporal(d).cpu().numpy()                 # [K, P, cells]
        for k, col in enumerate(cols):
            w = want(col).reshape(len(ob) - 1, -1)
            assert np.array_equal(np.isnan(got[k]), np.isnan(w)), (name, col)
            if col["inner"] == "sine_dd":
                np.testing.assert_allclose(got[k], w, rtol=1e-10, atol=1e-10, equal_nan=True)
            elif col.get("transform") == "pow":
                np.testing.assert_allclose(got[k], w, rtol=4e-15, equal_nan=True)
            else:
                np.testing.assert_array_equal(got[k], w)
        # the same plan with periods free to split over chunks: only the association of the outer sum changes
        fplan = hip.FusedPlan(T, ny * nx, hip.F64 if dtype == np.float64 else hip.F32, ib, ob, cols)
        if not any(c["outer"] in ("max", "min") or c.get("transform") == "hinge" for c in cols):
            assert int(fplan.describe().split("out_slots=")[1].split()[0]) > 3, fplan.describe()      # the long period was cut
        free = fplan.run_temporal(d).cpu().numpy()
        np.testing.assert_allclose(free, got, rtol=1e-12, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("glen", [4, 3])
@pytest.mark.parametrize("shape", [(150, 6, 20), (150, 5, 7), (151, 3, 130)])
def test_four_and_three_row_groups_take_the_lean_group_end(torch_cuda, dtype, shape, glen):
    """6-hourly / 8-hourly data: every inner group holds exactly four / three rows (`temporal.py:99-125` resampling 00/06/12/18
    or 00/08/16 UTC to the day).  Plans that qualify for the lean group end keep whole groups in registers (FEAT bit 10 / 12) instead
    of walking the generic row loop; the group sum adds the rows in time order, as the oracle does, and the mean is s * 0.25 (exact)
    / the correctly rounded s / 3 (`div_by` with the correctly rounded reciprocal), so mean / sum / min / max are bit-exact against
    `cport.block_stat`.  Plans that do not qualify (hinge, outer max, more columns than the variants hold) must leave the
    short-group path; unlike two-row groups, even one mean column takes it (it measured ahead of the ring, profiles/r03_quad_groups.txt)."""
    from aggfly_amd import hip
    T, ny, nx = glen * shape[0], shape[1], shape[2]
    rng = np.random.default_rng(23)
    k = np.arange(T)
    base = 15 + 10 * np.sin(2 * np.pi * (k // glen) / 365.0) + 6 * np.sin(2 * np.pi * (k % glen) / glen - np.pi / 2)
    cube = base[:, None, None] + rng.normal(0, 3, (T, ny, nx))
    cube[rng.integers(0, T, 9), rng.integers(0, ny, 9), rng.integers(0, nx, 9)] = np.nan
    cube[:, 1, 2] = np.nan
    cube = cube.astype(dtype)
    ib = np.arange(0, T + 1, glen, dtype=np.int64)
    G1 = T // glen
    ob = np.array([0, 20, 21, G1], dtype=np.int64)
    d = torch_cuda.from_numpy(cube).cuda()
    f64 = cube.astype(np.float64)
    code = hip.F64 if dtype == np.float64 else hip.F32
    suffix = "_quad" if glen == 4 else "_tri"

    def want(col):
        x = cport.resample(f64, ib, col["inner"], col.get("inner_args"))
        if col.get("transform") == "pow":
            x = cport.power(x, col["transform_arg"])
        elif col.get("transform") == "hinge":
            x = (x > col["transform_arg"]) * (x - col["transform_arg"])
        return cport.resample(x, ob, col["outer"])

    poly = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    mixed = [dict(inner="min", outer="sum"), dict(inner="max", transform="pow", transform_arg=2, outer="mean"),
             dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum"), dict(inner="mean", outer="mean"), dict(inner="sum", outer="sum"),
             dict(inner="sine_dd", inner_args=(0, 18, 1), outer="mean")]
    two = [dict(inner="max", outer="sum"), dict(inner="min", outer="mean")]
    generic = [dict(inner="mean", transform="hinge", transform_arg=20.0, outer="sum"), dict(inner="max", outer="max")]
    many = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4, 5, 6, 7)]
    light = [dict(inner="mean", outer="sum")]
    # (one- and two-column mean / sum plans on three-row float32 groups stream faster through the ring: the planner keeps them there)
    light_lean = not (glen == 3 and dtype == np.float32)
    for cols, quad in ((poly, True), (mixed, True), (two, True), (generic, False), (many, False), (light, light_lean)):
        plan = hip.FusedPlan(T, ny * nx, code, ib, ob, cols, exact_order=True)
        name = plan.describe().split()[0]
        assert name.endswith(suffix) == quad, name
        got = plan.run_temporal(d).cpu().numpy()
        for kk, col in enumerate(cols):
            w = want(col).reshape(len(ob) - 1, -1)
            assert np.array_equal(np.isnan(got[kk]), np.isnan(w)), (name, col)
            if col["inner"] == "sine_dd":
                np.testing.assert_allclose(got[kk], w, rtol=1e-10, atol=1e-10, equal_nan=True)
            elif col.get("transform") == "pow":
                np.testing.assert_allclose(got[kk], w, rtol=4e-15, equal_nan=True)
            else:
                np.testing.assert_array_equal(got[kk], w)
        fplan = hip.FusedPlan(T, ny * nx, code, ib, ob, cols)
        free = fplan.run_temporal(d).cpu().numpy()
        np.testing.assert_allclose(free, got, rtol=1e-12, atol=1e-12, equal_nan=True)
    # the knob that switches the form off gives the same numbers through the general path
    import os
    os.environ["AFHIP_NO_QUAD_MODE"] = "1"
    try:
        ring = hip.FusedPlan(T, ny * nx, code, ib, ob, poly, exact_order=True)
    finally:
        del os.environ["AFHIP_NO_QUAD_MODE"]
    assert suffix not in ring.describe().split()[0]
    quadp = hip.FusedPlan(T, ny * nx, code, ib, ob, poly, exact_order=True)
    assert quadp.describe().split()[0].endswith(suffix)
    np.testing.assert_allclose(quadp.run_temporal(d).cpu().numpy(), ring.run_temporal(d).cpu().numpy(), rtol=4e-15, equal_nan=True)


def _mixed_bounds(rng, G1, mix):
    """inner bounds with group lengths drawn from `mix` ({length: share}); the first and last groups are the mix's shortest / longest"""
    lens = rng.choice(list(mix), size=G1, p=np.array(list(mix.values()), dtype=float) / sum(mix.values()))
    lens[0], lens[-1] = min(mix), max(mix)
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mix", [{4: 0.9, 3: 0.07, 2: 0.03}, {1: 1, 2: 1, 3: 1, 4: 1}, {2: 0.5, 4: 0.5}, {1: 0.5, 2: 0.5}], ids=["gaps", "any", "2or4", "1or2"])
@pytest.mark.parametrize("shape", [(150, 6, 20), (151, 3, 130)])
def test_mixed_short_groups_take_the_lean_group_end(torch_cuda, monkeypatch, dtype, mix, shape):
    """A sub-daily series with missing steps (6-hourly data with gaps; a 12-hourly record joined to a 6-hourly one): inner groups of
    one to four rows, mixed.  `resample_groups` (`nb_kernels.py:80-115`) hands such bounds to the same kernels as any other; here the
    plans that qualify for the lean group end take the `_rag` form (FEAT bit 13: four row registers per group, a scalar length per
    group from the group table).  The sum runs in time order and the mean is the correctly rounded s / n, so mean / sum / min / max
    are bit-exact against `cport.block_stat` (`nb_kernels.py:121-155`); plans that do not qualify leave the short-group path; the
    knob that switches the form off gives the same numbers through the general path."""
    from aggfly_amd import hip
    G1, ny, nx = shape
    rng = np.random.default_rng(29 + G1)
    ib = _mixed_bounds(rng, G1, mix)
    T = int(ib[-1])
    k = np.arange(T)
    cube = (15 + 10 * np.sin(2 * np.pi * k / 365.0))[:, None, None] + rng.normal(0, 5, (T, ny, nx))
    cube[rng.integers(0, T, 9), rng.integers(0, ny, 9), rng.integers(0, nx, 9)] = np.nan
    cube[:, 1, 2] = np.nan
    cube = cube.astype(dtype)
    ob = np.array([0, 20, 21, G1], dtype=np.int64)
    d = torch_cuda.from_numpy(cube).cuda()
    f64 = cube.astype(np.float64)
    code = hip.F64 if dtype == np.float64 else hip.F32

    def want(col, obounds):
        x = cport.resample(f64, ib, col["inner"], col.get("inner_args"))
        if col.get("transform") == "pow":
            x = cport.power(x, col["transform_arg"])
        elif col.get("transform") == "hinge":
            x = (x > col["transform_arg"]) * (x - col["transform_arg"])
        return cport.resample(x, obounds, col["outer"])

    poly = [dict(inner="mean", transform="pow", transform_arg=e, outer="sum") for e in (1, 2, 3, 4)]
    mixed = [dict(inner="min", outer="sum"), dict(inner="max", transform="pow", transform_arg=2, outer="mean"),
             dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum"), dict(inner="mean", outer="mean"), dict(inner="sum", outer="sum"),
             dict(inner="sine_dd", inner_args=(0, 18, 1), outer="mean")]
    two = [dict(inner="max", outer="sum"), dict(inner="min", outer="mean")]
    generic = [dict(inner="mean", transform="hinge", transform_arg=20.0, outer="sum"), dict(inner="max", outer="max")]
    light = [dict(inner="mean", outer="sum")]
    # (one- and two-column mean / sum plans on float32 stream faster through the ring, as with three-row groups: the planner keeps them there)
    for cols, rag in ((poly, True), (mixed, True), (two, True), (generic, False), (light, dtype == np.float64)):
        plan = hip.FusedPlan(T, ny * nx, code, ib, ob, cols, exact_order=True)
        name = plan.describe().split()[0]
        assert name.endswith("_rag") == rag, name
        got = plan.run_temporal(d).cpu().numpy()
        for kk, col in enumerate(cols):
            w = want(col, ob).reshape(len(ob) - 1, -1)
            assert np.array_equal(np.isnan(got[kk]), np.isnan(w)), (name, col)
            if col["inner"] == "sine_dd":
                np.testing.assert_allclose(got[kk], w, rtol=1e-10, atol=1e-10, equal_nan=True)
            elif col.get("transform") == "pow":
                np.testing.assert_allclose(got[kk], w, rtol=4e-15, equal_nan=True)
            else:
                np.testing.assert_array_equal(got[kk], w)
    # many periods: one time chunk per period (chunks of a few groups, shorter than a block of rows; a chunk's first row is not the cube's)
    obm = np.unique(np.concatenate([[0], rng.integers(1, G1, 40), [G1]])).astype(np.int64)
    pm = hip.FusedPlan(T, ny * nx, code, ib, obm, mixed, exact_order=True)
    assert pm.describe().split()[0].endswith("_rag"), pm.describe()
    gm = pm.run_temporal(d).cpu().numpy()
    for kk, col in enumerate(mixed):
        w = want(col, obm).reshape(len(obm) - 1, -1)
        assert np.array_equal(np.isnan(gm[kk]), np.isnan(w)), col
        np.testing.assert_allclose(gm[kk], w, rtol=1e-10 if col["inner"] == "sine_dd" else 4e-15, atol=1e-10 if col["inner"] == "sine_dd" else 0, equal_nan=True)
    monkeypatch.setenv("AFHIP_NO_RAGGED_MODE", "1")
    ring = hip.FusedPlan(T, ny * nx, code, ib, ob, poly, exact_order=True)
    monkeypatch.delenv("AFHIP_NO_RAGGED_MODE")
    assert "_rag" not in ring.describe().split()[0] and "_pair" not in ring.describe().split()[0], ring.describe()
    ragp = hip.FusedPlan(T, ny * nx, code, ib, ob, poly, exact_order=True)
    np.testing.assert_allclose(ragp.run_temporal(d).cpu().numpy(), ring.run_temporal(d).cpu().numpy(), rtol=4e-15, equal_nan=True)
    # an empty group among them, or one of five rows: not a short-group plan
    for bad in (np.concatenate([ib[:5], ib[4:]]), np.concatenate([ib[:3], ib[3:] + 4])):
        Tb = int(bad[-1])
        pb = hip.FusedPlan(Tb, ny * nx, code, bad, np.array([0, len(bad) - 1], dtype=np.int64), poly)
        assert "_rag" not in pb.describe().split()[0], pb.describe()


@pytest.mark.parametrize("glen", [2, 3, 4])
@pytest.mark.parametrize("ngroups", [1, 2, 3, 5, 9])
def test_short_group_plans_shorter_than_a_block_of_rows(torch_cuda, glen, ngroups):
    """Chunks that hold fewer rows than the short-group forms keep in flight (8): the prologue loads only the rows that exist,
    the group loop stops at the chunk's last group, and no request for "the next block" is issued past the end."""
    from aggfly_amd import hip
    T, ny, nx = glen * ngroups, 3, 70
    rng = np.random.default_rng(5 + ngroups)
    cube = (15 + rng.normal(0, 8, (T, ny, nx))).astype(np.float32)
    cube[0, 0, 0] = np.nan
    ib = np.arange(0, T + 1, glen, dtype=np.int64)
    ob = np.array([0, ngroups], dtype=np.int64) if ngroups < 3 else np.array([0, 1, ngroups], dtype=np.int64)
    cols = [dict(inner="max", outer="sum"), dict(inner="mean", transform="pow", transform_arg=3, outer="sum"),
            dict(inner="sine_dd", inner_args=(10, 30, 0), outer="sum")]
    d = torch_cuda.from_numpy(cube).cuda()
    f64 = cube.astype(np.float64)
    for dtype, code, dev in ((np.float32, hip.F32, d), (np.float64, hip.F64, d.double())):
        plan = hip.FusedPlan(T, ny * nx, code, ib, ob, cols, exact_order=True)
        assert "_pair" in plan.describe().split()[0], plan.describe()
        got = plan.run_temporal(dev).cpu().numpy()
        for k, col in enumerate(cols):
            x = cport.resample(f64, ib, col["inner"], col.get("inner_args"))
            if col.get("transform") == "pow":
                x = cport.power(x, col["transform_arg"])
            w = cport.resample(x, ob, col["outer"]).reshape(len(ob) - 1, -1)
            assert np.array_equal(np.isnan(got[k]), np.isnan(w))
            np.testing.assert_allclose(got[k], w, rtol=1e-10 if col["inner"] == "sine_dd" else 4e-15, atol=1e-10 if col["inner"] == "sine_dd" else 0, equal_nan=True)


def test_read_probe_and_build_info(torch_cuda):
    """The measuring aids of the ABI: `afhip_read_probe` times a bare streaming read of a cube (bench.py's measured read ceiling) and
    refuses what it cannot stream; `afhip_build_info` names the menu the library was built from."""
    from aggfly_amd import hip
    torch = torch_cuda
    cube = torch.zeros((512, 64, 256), dtype=torch.float32, device="cuda")       # 512 rows of 64 KiB
    ms = hip.read_probe(cube, 5)
    assert len(ms) == 5 and all(0.0 < m < 50.0 for m in ms), ms
    gbps = cube.numel() * 4 / (min(ms) * 1e-3) / 1e9
    assert gbps > 50.0, gbps                                                      # (a 33 MB cube: cache-resident, far below the HBM figure's scale)
    odd = torch.zeros((8, 3), dtype=torch.float32, device="cuda")                # rows of 12 bytes: not a multiple of 8
    with pytest.raises(ValueError, match="multiples of 8"):
        hip.read_probe(odd, 2)
    info = hip.build_info()
    assert info["abi"] == hip.ABI_VERSION and info["variants"] > 300 and info["region_fused_twins"] > 50
