"""N > 1 path on CPU: world_size-2 gloo process groups exercise the sharding maths and the
two exchange steps (panel all_gather for time shards, num/den all_reduce for cell shards).
The per-shard numbers come from the oracle, so what is checked is exactly that
"shard -> local reduce -> exchange" equals the unsharded result."""
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aggfly_amd import distributed as D, synth
from aggfly_amd.timegroups import resample_groups
from oracle import cport
from oracle.ref_spatial import spatial_num_den


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _case():
    T, ny, nx = 24 * 80, 8, 10
    cube = synth.temperature_cube(T, ny, nx, seed=31, ocean_frac=0.1, scattered_nan=25)
    time = pd.date_range("2003-01-10", periods=T, freq="h")
    tab = synth.weights_table(ny, nx, 6, seed=32, secondary=True)
    return cube, time, tab, ny, nx


def _cells(cube, time):
    """[K=2, P, cells]: mean@date -> pow(1,2) -> sum@month via the oracle's C port."""
    ib, lab = resample_groups(time, "1D")
    ob, labels = resample_groups(lab, "ME")
    m = cport.resample(cube, ib, "mean")
    return np.stack([cport.resample(np.power(m, e), ob, "sum").reshape(len(labels), -1) for e in (1, 2)]), labels


def _full_res(cube, time, tab, ncells):
    cells, labels = _cells(cube, time)
    nums, den, _ = spatial_num_den({f"k{k}": cells[k].T for k in range(2)}, tab, np.arange(ncells))
    num = np.stack([nums["k0"], nums["k1"]])
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(den[None] != 0, num / den[None], np.nan), labels


def _seasonal_case():
    """Daily June-August data of 2000-2002: the months September-May are empty resample bins, and with three ranks the
    share boundaries (27 monthly periods -> 9 each) fall INSIDE those gaps: rank 1's share runs 2001-03 .. 2001-11 but
    its data only 2001-06 .. 2001-08."""
    days = pd.DatetimeIndex(np.concatenate([pd.date_range(f"{y}-06-01", f"{y}-08-31", freq="D") for y in (2000, 2001, 2002)]))
    ny, nx = 6, 7
    cube = synth.temperature_cube(len(days), ny, nx, seed=41, ocean_frac=0.1, scattered_nan=12)
    tab = synth.weights_table(ny, nx, 5, seed=42, secondary=True)
    return cube, days, tab, ny, nx


def _worker(rank, ws, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        if mode == "time_gaps":
            cube, time, tab, ny, nx = _seasonal_case()
            want, labels = _full_res(cube, time, tab, ny * nx)
            k_lo, k_hi, p_lo, p_hi, P = D.time_shard_bounds(time, "ME", rank, ws)
            assert P == len(labels) == 27
            counts = [D.split_even(P, r, ws)[1] - D.split_even(P, r, ws)[0] for r in range(ws)]
            local, local_labels = _full_res(cube[k_lo:k_hi], time[k_lo:k_hi], tab, ny * nx)
            assert local.shape[2] < p_hi - p_lo                      # leading / trailing empty bins are not there locally
            block = D.place_by_label(torch.from_numpy(local), local_labels, labels, p_lo, p_hi)
            full = D.gather_panel(block, counts).numpy()
            np.testing.assert_array_equal(full, want)
            assert np.isnan(want[:, :, 3:12]).all() and not np.isnan(want[:, :, 12:15]).all()      # Sept-May empty, JJA 2001 there
            with pytest.raises(ValueError, match="periods"):        # the unplaced block is refused, not shifted
                D.gather_panel(torch.from_numpy(local), counts)
            q.put((rank, "ok"))
            return
        cube, time, tab, ny, nx = _case()
        want, labels = _full_res(cube, time, tab, ny * nx)
        if mode == "time":
            k_lo, k_hi, p_lo, p_hi, P = D.time_shard_bounds(time, "ME", rank, ws)
            assert P == len(labels)
            local, _ = _full_res(cube[k_lo:k_hi], time[k_lo:k_hi], tab, ny * nx)
            assert local.shape[2] == p_hi - p_lo
            counts = [D.split_even(P, r, ws)[1] - D.split_even(P, r, ws)[0] for r in range(ws)]
            full = D.gather_panel(torch.from_numpy(local), counts).numpy()
            np.testing.assert_array_equal(full, want)             # periods never straddle ranks: bit-identical
        else:
            y0, y1 = D.split_even(ny, rank, ws)
            cells, _ = _cells(cube, time)
            band = cells.reshape(2, -1, ny, nx)[:, :, y0:y1, :].reshape(2, cells.shape[1], -1)
            rows, cols, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
            br, bc, bw = D.band_csr_triplets(rows, cols, w, ny, nx, y0, y1)
            btab = pd.DataFrame({"index_right": br, "cell_id": bc, "weight": bw})
            R = int(rows.max()) + 1
            num = np.zeros((2, R, cells.shape[1])); den = np.zeros((R, cells.shape[1]))
            if len(btab):
                nums, d, ids = spatial_num_den({f"k{k}": band[k].T for k in range(2)}, btab, np.arange((y1 - y0) * nx))
                num[:, ids] = np.stack([nums["k0"], nums["k1"]]); den[ids] = d
            _, _, res = D.reduce_num_den(torch.from_numpy(num), torch.from_numpy(den))
            np.testing.assert_allclose(res.numpy(), want, rtol=1e-13, equal_nan=True)
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["time", "cells"])
def test_world2_exchange(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(out) == [(0, "ok"), (1, "ok")], out


def test_world3_time_shards_with_gaps_on_the_boundaries():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, "time_gaps", q)) for r in range(3)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(out) == [(0, "ok"), (1, "ok"), (2, "ok")], out


# ----------------------------------------------------------------------------------------------------------------------
# world_size 8: the BASELINE splits as the 8-GPU node will see them, rehearsed on gloo with oracle cells
# ----------------------------------------------------------------------------------------------------------------------
def _case8(name):
    """-> (cube, time, tab, ny, nx, inner freq, outer freq, periods expected)"""
    from aggfly_amd.cfcalendar import cf_range
    if name == "years40":            # configs[2]: 40 annual periods, five per rank
        ny, nx, time = 3, 4, pd.date_range("1980-01-01", "2019-12-31", freq="D")
        return ny, nx, time, "1D", "YE", 40
    if name == "years251_noleap":    # configs[3]: 251 noleap years -> 32,32,32,31,31,31,31,31: the padded gather
        ny, nx, time = 3, 4, cf_range("1850-01-01", 251 * 365, "D", "noleap")
        return ny, nx, time, "1D", "YE", 251
    if name == "months12":           # the reference's benchmark shape: 12 months -> 2,2,2,2,1,1,1,1
        ny, nx, time = 4, 5, pd.date_range("2001-01-01", periods=8760, freq="h")
        return ny, nx, time, "1D", "ME", 12
    if name == "years5":             # fewer periods than ranks: three ranks hold no period at all
        ny, nx, time = 3, 4, pd.date_range("2001-01-01", "2005-12-31", freq="D")
        return ny, nx, time, "1D", "YE", 5
    if name == "bands215":           # configs[0], [1], [4]: ONE period -> the counties extent's 215 grid rows in 8 bands
        ny, nx, time = 215, 6, pd.date_range("2001-01-01", periods=72, freq="h")
        return ny, nx, time, "1D", "YE", 1
    raise KeyError(name)


def _res8(cube, time, tab, ncells, inner, outer):
    """res[K=2, R, P] of mean@inner -> pow(1, 2) -> sum@outer + the weighted average, all by the oracle; an empty share -> P = 0."""
    R = int(tab["index_right"].max()) + 1
    if len(time) == 0:
        return np.empty((2, R, 0)), time
    ib, lab = resample_groups(time, inner)
    ob, labels = resample_groups(lab, outer)
    m = cport.resample(cube, ib, "mean")
    cells = np.stack([cport.resample(np.power(m, e), ob, "sum").reshape(len(labels), -1) for e in (1, 2)])
    nums, den, ids = spatial_num_den({f"k{k}": cells[k].T for k in range(2)}, tab, np.arange(ncells))
    num = np.zeros((2, R, len(labels))); d = np.zeros((R, len(labels)))
    num[:, ids] = np.stack([nums["k0"], nums["k1"]]); d[ids] = den
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(d[None] != 0, num / d[None], np.nan), labels, num, d


def _worker8(rank, ws, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        ny, nx, time, inner, outer, P_want = _case8(name)
        cube = synth.temperature_cube(len(time), ny, nx, seed=71, ocean_frac=0.1, scattered_nan=20)
        tab = synth.weights_table(ny, nx, 7, seed=72, secondary=True)
        want, labels, _, _ = _res8(cube, time, tab, ny * nx, inner, outer)
        assert want.shape[2] == P_want
        if name == "bands215":
            y0, y1 = D.split_even(ny, rank, ws)
            assert y1 - y0 == (27 if rank < 7 else 26)
            rows, cols, w = tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy()
            br, bc, bw = D.band_csr_triplets(rows, cols, w, ny, nx, y0, y1)
            R = int(rows.max()) + 1
            num, den = np.zeros((2, R, 1)), np.zeros((R, 1))
            if len(br):
                btab = pd.DataFrame({"index_right": br, "cell_id": bc, "weight": bw})
                _, _, n_b, d_b = _res8(np.ascontiguousarray(cube[:, y0:y1, :]), time, btab, (y1 - y0) * nx, inner, outer)
                num[:, :n_b.shape[1]] = n_b; den[:d_b.shape[0]] = d_b
            _, _, res = D.reduce_num_den(torch.from_numpy(num), torch.from_numpy(den))
            np.testing.assert_allclose(res.numpy(), want, rtol=1e-13, equal_nan=True)
        else:
            k_lo, k_hi, p_lo, p_hi, P = D.time_shard_bounds(time, outer, rank, ws)
            assert P == P_want
            counts = [D.split_even(P, r, ws)[1] - D.split_even(P, r, ws)[0] for r in range(ws)]
            if name == "years251_noleap":
                assert counts == [32, 32, 32, 31, 31, 31, 31, 31] and k_hi - k_lo == counts[rank] * 365
            if name == "months12":
                assert counts == [2, 2, 2, 2, 1, 1, 1, 1]
            if name == "years5":
                assert counts == [1, 1, 1, 1, 1, 0, 0, 0] and (k_hi > k_lo) == (rank < 5)
            local = _res8(cube[k_lo:k_hi], time[k_lo:k_hi], tab, ny * nx, inner, outer)
            block = D.place_by_label(torch.from_numpy(local[0]), local[1] if len(local[1]) else labels[:0], labels, p_lo, p_hi)
            full = D.gather_panel(block, counts).numpy()
            np.testing.assert_array_equal(full, want)                 # whole periods per rank: bit-identical, padding trimmed
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-800:]}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["years40", "years251_noleap", "months12", "years5", "bands215"])
def test_world8_baseline_splits(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, name, q)) for r in range(8)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(out) == [(r, "ok") for r in range(8)], out


def test_place_by_label():
    from aggfly_amd.cfcalendar import cf_range
    labels = pd.date_range("2001-01-31", periods=12, freq="ME")
    res = torch.arange(2 * 3 * 3, dtype=torch.float64).reshape(2, 3, 3)
    blk = D.place_by_label(res, labels[5:8], labels, 2, 11)
    assert blk.shape == (2, 3, 9) and torch.equal(blk[:, :, 3:6], res)
    assert torch.isnan(blk[:, :, :3]).all() and torch.isnan(blk[:, :, 6:]).all()
    assert D.place_by_label(res, labels[2:5], labels, 2, 5) is res               # already in place: no copy
    empty = D.place_by_label(res[:, :, :0], labels[:0], labels, 4, 6)
    assert empty.shape == (2, 3, 2) and torch.isnan(empty).all()
    with pytest.raises(ValueError, match="outside its own share"):
        D.place_by_label(res, labels[5:8], labels, 6, 11)
    with pytest.raises(ValueError, match="does not have"):
        D.place_by_label(res, pd.date_range("2001-01-15", periods=3, freq="D"), labels, 0, 12)
    _, cf = resample_groups(cf_range("2001-01-01", 720, "D", "360_day"), "ME")
    blk = D.place_by_label(res, cf[7:10], cf, 6, 12)
    assert blk.shape == (2, 3, 6) and torch.equal(blk[:, :, 1:4], res)
    assert list(D.label_positions(cf[3:5], cf)) == [3, 4]


def test_shard_maths_single_process():
    assert [D.split_even(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [D.split_even(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    spec = dict(a=[("aggregate", {"calc": "mean", "groupby": "date"}), ("aggregate", {"calc": "sum", "groupby": "year"})],
                b=[("aggregate", {"calc": "max", "groupby": "year"})])
    assert D.output_freq(spec) == "YE"
    with pytest.raises(ValueError, match="share one output frequency"):
        D.output_freq(dict(spec, c=[("aggregate", {"calc": "max", "groupby": "month"})]))
    t = pd.date_range("1999-12-31 12:00", periods=24 * 800, freq="h")
    spans = [D.time_shard_bounds(t, "YE", r, 3) for r in range(3)]
    assert spans[0][0] == 0 and spans[-1][1] == len(t) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert D.world() == (0, 1)


def test_plan_windows_cuts_on_output_periods():
    """`distributed.plan_windows`: a rank's output periods are taken in consecutive runs whose time steps fit the
    HBM budget; a period larger than the budget still gets a window of its own."""
    import numpy as np
    from aggfly_amd.distributed import plan_windows
    b = np.array([0, 10, 20, 30, 40, 55])
    assert plan_windows(b, 0, 5, 8, 160) == [(0, 2), (2, 4), (4, 5)]
    assert plan_windows(b, 0, 5, 8, 1) == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]
    assert plan_windows(b, 1, 4, 8, 10 ** 9) == [(1, 4)]
    assert plan_windows(b, 2, 2, 8, 100) == []
    assert plan_windows(b, 0, 5, None, 100) == [(0, 5)]
    for budget in (1, 80, 81, 239, 240, 10 ** 6):
        runs = plan_windows(b, 0, 5, 8, budget)
        assert runs[0][0] == 0 and runs[-1][1] == 5 and all(x[1] == y[0] for x, y in zip(runs, runs[1:]))
        assert all((b[hi] - b[lo]) * 8 <= budget or hi == lo + 1 for lo, hi in runs)
