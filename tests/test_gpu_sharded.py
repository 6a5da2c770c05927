"""The N > 1 path on the GPU box.  Every job script selects its rank's card and its backend the way the CLI does
(`distributed.init_local_rank`): with a GPU per rank the ranks sit on DISTINCT cards and exchange over RCCL ("nccl"); on a
one-GPU box they share card 0 and rehearse the same code over gloo.  Each job reports which it was and the tests assert it
(`_check_backend`), so the day these tests run on a multi-GPU node they measure RCCL, not card 0."""
import os
import socket
import subprocess
import sys

import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
import numpy as np, pandas as pd, torch, torch.distributed as dist
import aggfly_amd as af
from aggfly_amd import synth, distributed as D

mode, out = sys.argv[1], sys.argv[2]
# one process per GPU: this rank's card and the backend are chosen exactly as the CLI does (aggfly_amd/cli/main.py):
# RCCL when the node has a GPU per local rank, a gloo rehearsal on one card otherwise; the test reads back which it was
info = D.init_local_rank(os.environ.get("AGGFLY_DIST_BACKEND"))
T, ny, nx = 24 * 400, 10, 12
cube = synth.temperature_cube(T, ny, nx, seed=61, ocean_frac=0.1, scattered_nan=30)
time = pd.date_range("2003-03-01", periods=T, freq="h")
ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                             {"time": time, "latitude": 30 + 0.25 * np.arange(ny), "longitude": 250 + 0.25 * np.arange(nx)}), lon_is_360=True)
tab = synth.weights_table(ny, nx, 6, seed=62, secondary=True)
gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
w = af.weights_from_objects(ds, gr, table=tab)
freq = "month" if mode == "time" else "year"
spec = dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": freq})],
            t=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
               ("aggregate", {"calc": "sum", "groupby": freq})])
if dist.is_initialized():
    df = D.aggregate_dataset_sharded(w, ds, spec, shard=mode)
else:
    df = af.aggregate_dataset(dataset=ds, weights=w, aggregator_dict=spec)
used = D.devices_used()
if D.world()[0] == 0:
    df.to_csv(out, index=False)
    import json
    json.dump(dict(info, **used), open(out + ".dist.json", "w"))
if dist.is_initialized():
    dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _check_backend(out_csv, ranks):
    """The job wrote which backend and which cards its ranks used: with a GPU per rank the exchange must have gone over RCCL
    between `ranks` DISTINCT devices; on fewer cards it is a gloo rehearsal (ranks share card 0 — the only way this
    one-GPU box can run the N > 1 code path at all)."""
    import json
    import torch
    d = json.load(open(out_csv + ".dist.json"))
    ndev = torch.cuda.device_count()
    assert d["ranks"] == ranks and d["devices_visible"] == ndev
    if os.environ.get("AGGFLY_DIST_BACKEND"):
        assert d["backend"] == os.environ["AGGFLY_DIST_BACKEND"]
    elif ndev >= ranks:
        assert d["backend"] == "nccl" and d["devices_used"] == ranks, d
    else:
        assert d["backend"] == "gloo" and d["devices_used"] == min(ndev, ranks), d
    return d


@pytest.mark.parametrize("mode", ["time", "cells"])
def test_sharded_equals_single(torch_cuda, tmp_path, mode):
    script = tmp_path / "job.py"
    script.write_text(SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # with exact_order no period is ever split, so a period's sums do not depend on which shard
    # (or how long a cube) it is computed in: the time-sharded panel is bit-identical
    env = dict(os.environ, PYTHONPATH=root, AGGFLY_HIP_EXACT_ORDER="1")
    one, two = str(tmp_path / "one.csv"), str(tmp_path / "two.csv")
    r = subprocess.run([sys.executable, str(script), mode, one], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script), mode, two], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_backend(two, 2)
    a, b = pd.read_csv(one), pd.read_csv(two)
    assert list(a.columns) == list(b.columns) and len(a) == len(b) > 0
    pd.testing.assert_frame_equal(a, b, rtol=1e-12 if mode == "cells" else 0, atol=0, check_exact=(mode == "time"))


STORE_SCRIPT = r'''
import os, sys
import numpy as np, pandas as pd, torch, torch.distributed as dist
import aggfly_amd as af
from aggfly_amd import synth, distributed as D

store, out = sys.argv[1], sys.argv[2]
# one process per GPU: this rank's card and the backend are chosen exactly as the CLI does (aggfly_amd/cli/main.py):
# RCCL when the node has a GPU per local rank, a gloo rehearsal on one card otherwise; the test reads back which it was
info = D.init_local_rank(os.environ.get("AGGFLY_DIST_BACKEND"))
ny, nx = 10, 12
tab = synth.weights_table(ny, nx, 6, seed=62, secondary=True)
gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
spec = dict(dd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}), ("aggregate", {"calc": "sum", "groupby": "month"})],
            t=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
               ("aggregate", {"calc": "sum", "groupby": "month"})])
weights_of = lambda ds: af.weights_from_objects(ds, gr, table=tab)
if dist.is_initialized() and os.environ.get("SHARD") == "cells":
    # latitude bands: each rank decodes only the chunks that touch its band, one all_reduce finishes the panel
    from aggfly_amd import codec
    seen = []
    real = codec.decode_ranges
    codec.decode_ranges = lambda kind, locs, outs, threads=8: seen.append(len(locs)) or real(kind, locs, outs, threads)
    df = D.aggregate_store_cells(weights_of, store, "t2m", spec, lon_is_360=True, preprocess=lambda x: x - 273.15)
    assert 0 < sum(seen) <= int(os.environ["CHUNKS_PER_RANK"]), seen
elif dist.is_initialized():
    from aggfly_amd import codec
    seen = []
    real = codec.decode_ranges
    codec.decode_ranges = lambda kind, locs, outs, threads=8: seen.append(len(locs)) or real(kind, locs, outs, threads)
    df = D.aggregate_store_sharded(weights_of, store, "t2m", spec, lon_is_360=True, preprocess=lambda x: x - 273.15)
    assert 0 < sum(seen) < 20, seen          # each rank decoded about half of the store's 28 chunks, not all of them
elif os.environ.get("WINDOW_BYTES"):
    # one GPU, a budget of a few output periods per window: the store goes through HBM window by window
    from aggfly_amd import io as afio
    opened = []
    real_open = afio.dataset_from_path
    def spy(*a, **k):
        opened.append(k.get("time_window"))
        return real_open(*a, **k)
    afio.dataset_from_path = spy
    df = D.aggregate_store_sharded(weights_of, store, "t2m", spec, lon_is_360=True, preprocess=lambda x: x - 273.15,
                                   max_window_bytes=int(os.environ["WINDOW_BYTES"]))
    assert len(opened) >= 4 and opened[0][0] == 0 and opened[-1][1] == 24 * 400, opened
    assert all(a[1] == b[0] for a, b in zip(opened, opened[1:])), opened
else:
    ds = af.dataset_from_path(store, "t2m", lon_is_360=True, preprocess=lambda x: x - 273.15, device="cuda")
    df = af.aggregate_dataset(dataset=ds, weights=weights_of(ds), aggregator_dict=spec)
used = D.devices_used()
if D.world()[0] == 0:
    df.to_csv(out, index=False)
    import json
    json.dump(dict(info, **used), open(out + ".dist.json", "w"))
if dist.is_initialized():
    dist.destroy_process_group()
'''


def test_store_sharded_streams_each_ranks_window(torch_cuda, tmp_path):
    """`aggregate_store_sharded`: two ranks open the same Zarr store, each streams only the time steps of its own
    output periods into the GPU, and the gathered panel equals the single-process run."""
    import numpy as np
    import aggfly_amd as af
    from aggfly_amd import synth
    T, ny, nx = 24 * 400, 10, 12
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=61, ocean_frac=0.1, scattered_nan=30) + np.float32(273.15)
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                                 {"time": pd.date_range("2003-03-01", periods=T, freq="h"), "latitude": 30 + 0.25 * np.arange(ny),
                                  "longitude": 250 + 0.25 * np.arange(nx)}), lon_is_360=True)
    store = str(tmp_path / "era.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 350, "latitude": ny, "longitude": nx})
    script = tmp_path / "store_job.py"
    script.write_text(STORE_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, AGGFLY_HIP_EXACT_ORDER="1")
    one, two = str(tmp_path / "one.csv"), str(tmp_path / "two.csv")
    r = subprocess.run([sys.executable, str(script), store, one], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script), store, two], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_backend(two, 2)
    a, b = pd.read_csv(one), pd.read_csv(two)
    assert len(a) == len(b) > 0
    pd.testing.assert_frame_equal(a, b, check_exact=True)
    # a store longer than the HBM budget: windows of whole output periods (here 4 months of hourly steps at most)
    win = str(tmp_path / "win.csv")
    r = subprocess.run([sys.executable, str(script), store, win], capture_output=True, text=True, timeout=300,
                       env=dict(env, WINDOW_BYTES=str(4 * 31 * 24 * ny * nx * 4)))
    assert r.returncode == 0, r.stderr[-2000:]
    pd.testing.assert_frame_equal(a, pd.read_csv(win), check_exact=True)


def test_store_cell_sharding_reads_only_its_band(torch_cuda, tmp_path):
    """`aggregate_store_cells`: two ranks take latitude bands of the same store (space-tiled chunks: each rank decodes
    only the tiles of its band), the numerators / denominators are all-reduced, and the frame equals the
    single-process run to rounding (band sums are added in a different order)."""
    import numpy as np
    import aggfly_amd as af
    from aggfly_amd import synth
    T, ny, nx = 24 * 90, 10, 12
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=63, ocean_frac=0.1, scattered_nan=30) + np.float32(273.15)
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                                 {"time": pd.date_range("2003-03-01", periods=T, freq="h"), "latitude": 30 + 0.25 * np.arange(ny),
                                  "longitude": 250 + 0.25 * np.arange(nx)}), lon_is_360=True)
    store = str(tmp_path / "era.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 720, "latitude": 5, "longitude": 6})      # 3 x 2 x 2 tiles
    script = tmp_path / "store_job.py"
    script.write_text(STORE_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    one, two = str(tmp_path / "one.csv"), str(tmp_path / "two.csv")
    r = subprocess.run([sys.executable, str(script), store, one], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script), store, two], capture_output=True, text=True, timeout=300,
                       env=dict(env, SHARD="cells", CHUNKS_PER_RANK="6"))
    assert r.returncode == 0, r.stderr[-2000:]
    _check_backend(two, 2)
    a, b = pd.read_csv(one), pd.read_csv(two)
    assert list(a.columns) == list(b.columns) and len(a) == len(b) > 0
    pd.testing.assert_frame_equal(a, b, rtol=1e-12, atol=0)


SEASONAL_SCRIPT = r'''
import os, sys
import numpy as np, pandas as pd, torch, torch.distributed as dist
import aggfly_amd as af
from aggfly_amd import synth, distributed as D

store, out = sys.argv[1], sys.argv[2]
# one process per GPU: this rank's card and the backend are chosen exactly as the CLI does (aggfly_amd/cli/main.py):
# RCCL when the node has a GPU per local rank, a gloo rehearsal on one card otherwise; the test reads back which it was
info = D.init_local_rank(os.environ.get("AGGFLY_DIST_BACKEND"))
ny, nx = 6, 8
tab = synth.weights_table(ny, nx, 5, seed=72, secondary=True)
gr = af.GeoRegions(pd.DataFrame({"geoid": [f"r{i}" for i in range(int(tab.index_right.max()) + 1)]}))
spec = dict(t=[("aggregate", {"calc": "mean", "groupby": "date"}), ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
               ("aggregate", {"calc": "sum", "groupby": "month"})],
            hot=[("aggregate", {"calc": "bins", "groupby": "month", "ddargs": [25, 99, 0]})])
weights_of = lambda ds: af.weights_from_objects(ds, gr, table=tab)
if dist.is_initialized() or os.environ.get("WINDOW_BYTES"):
    wb = os.environ.get("WINDOW_BYTES")
    df = D.aggregate_store_sharded(weights_of, store, "t2m", spec, lon_is_360=True, max_window_bytes=int(wb) if wb else None)
else:
    ds = af.dataset_from_path(store, "t2m", lon_is_360=True, device="cuda")
    df = af.aggregate_dataset(dataset=ds, weights=weights_of(ds), aggregator_dict=spec)
if os.environ.get("RESIDENT_SHARDED"):
    ds = af.dataset_from_path(store, "t2m", lon_is_360=True, device="cuda")
    df2 = D.aggregate_dataset_sharded(weights_of(ds), ds, spec, shard="time")
    pd.testing.assert_frame_equal(df, df2, check_exact=True)
used = D.devices_used()
if D.world()[0] == 0:
    df.to_csv(out, index=False)
    import json
    json.dump(dict(info, **used), open(out + ".dist.json", "w"))
if dist.is_initialized():
    dist.destroy_process_group()
'''


def test_seasonal_store_gaps_on_shard_and_window_boundaries(torch_cuda, tmp_path):
    """Daily June-August data of three years: the nine months between the seasons are empty resample bins.  Shares of
    three ranks and HBM windows then begin or end INSIDE a gap, where the local time slice has fewer periods than the
    share: results must be placed by label (`distributed.place_by_label`), never by position."""
    import numpy as np
    import aggfly_amd as af
    from aggfly_amd import synth
    days = pd.DatetimeIndex(np.concatenate([pd.date_range(f"{y}-06-01", f"{y}-08-31", freq="D") for y in (2000, 2001, 2002)]))
    ny, nx = 6, 8
    cube = synth.temperature_cube(len(days), ny, nx, dtype=np.float32, seed=71, ocean_frac=0.1, scattered_nan=10)
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                                 {"time": days, "latitude": 30 + 0.25 * np.arange(ny), "longitude": 250 + 0.25 * np.arange(nx)}), lon_is_360=True)
    store = str(tmp_path / "jja.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 40, "latitude": ny, "longitude": nx})
    script = tmp_path / "seasonal_job.py"
    script.write_text(SEASONAL_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, AGGFLY_HIP_EXACT_ORDER="1")
    one, three, win, both = (str(tmp_path / f"{n}.csv") for n in ("one", "three", "win", "both"))
    r = subprocess.run([sys.executable, str(script), store, one], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    a = pd.read_csv(one)
    assert len(a) > 0 and a["time"].nunique() == 9                                   # the 18 empty months are dropped rows
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1"]
    r = subprocess.run(launch + ["--master-port", str(_free_port()), str(script), store, three], capture_output=True, text=True,
                       timeout=300, env=dict(env, RESIDENT_SHARDED="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    _check_backend(three, 3)
    pd.testing.assert_frame_equal(a, pd.read_csv(three), check_exact=True)
    # windows of at most 40 days of steps: (June), (July), (August + the nine empty months), ...
    wb = str(40 * ny * nx * 4)
    r = subprocess.run([sys.executable, str(script), store, win], capture_output=True, text=True, timeout=300, env=dict(env, WINDOW_BYTES=wb))
    assert r.returncode == 0, r.stderr[-2000:]
    pd.testing.assert_frame_equal(a, pd.read_csv(win), check_exact=True)
    r = subprocess.run(launch + ["--master-port", str(_free_port()), str(script), store, both], capture_output=True, text=True,
                       timeout=300, env=dict(env, WINDOW_BYTES=wb))
    assert r.returncode == 0, r.stderr[-2000:]
    pd.testing.assert_frame_equal(a, pd.read_csv(both), check_exact=True)


def test_bench_two_rank_rehearsal_reports_backend_and_devices(torch_cuda, tmp_path):
    """bench.py with 2 ranks on this one-GPU box: refused by default; with AGGFLY_BENCH_BACKEND=gloo it runs as a labelled
    rehearsal — the JSON line says backend gloo, devices_visible 1, devices_used 1, and carries the warning.  Both sharding arms."""
    import json
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU visible: the rehearsal path is for one-GPU boxes")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1"]
    small = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--ny", "40", "--nx", "64", "--T", "720", "--regions", "30"]
    env = {k: v for k, v in os.environ.items() if k not in ("AGGFLY_BENCH_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(launch + ["--master-port", str(_free_port()), bench] + small, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "RCCL needs one GPU per rank" in r.stderr
    # the driver's own command line — `python bench.py --gpus 2`, no torchrun around it — is refused the same way, in one line ...
    r = subprocess.run([sys.executable, bench] + small, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "RCCL needs one GPU per rank" in r.stderr and "Traceback" not in r.stderr
    for shard, scaling in (("time", "weak"), ("cells", "strong")):
        # ... and as a rehearsal starts its own two ranks (time arm) / runs under an outer torchrun (cell arm): same line either way
        cmd = ([sys.executable, bench] if shard == "time" else launch + ["--master-port", str(_free_port()), bench]) + small + ["--shard", shard]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, AGGFLY_BENCH_BACKEND="gloo"))
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["backend"] == "gloo" and line["devices_visible"] == 1 and line["devices_used"] == 1 and line["ranks"] == 2
        assert line["n_gpus"] == 2 and line["scaling"] == scaling and "warning" in line and "gloo" in line["config"]["sharding"]
        assert line["value"] > 0 and line["roofline"]["frac"] > 0
    # one process, cell arm: the divide after the (absent) exchange reproduces the plan's own panel
    r = subprocess.run([sys.executable, bench, "--steps", "2", "--warmup", "1", "--ny", "40", "--nx", "64", "--T", "720", "--regions", "30",
                        "--shard", "cells", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["backend"].startswith("none") and line["devices_used"] == 1 and line["scaling"] == "strong"


def test_bench_two_ranks_on_two_gpus_use_rccl(torch_cuda, tmp_path):
    """With two (or more) GPUs visible, `bench.py --gpus 2` must run on RCCL with one card per rank in both sharding arms:
    backend nccl, two distinct devices, no rehearsal warning.  (Skipped on one-GPU boxes, where the test above covers the
    refusal and the labelled gloo rehearsal.)"""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    small = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--ny", "40", "--nx", "64", "--T", "720", "--regions", "30"]
    env = {k: v for k, v in os.environ.items() if k not in ("AGGFLY_BENCH_BACKEND", "AGGFLY_DIST_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for shard, scaling in (("time", "weak"), ("cells", "strong")):
        # the plain command the driver runs: bench.py starts its own ranks (a child torchrun)
        r = subprocess.run([sys.executable, bench] + small + ["--shard", shard], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["backend"] == "nccl" and line["ranks"] == 2 and line["devices_used"] == 2 and "warning" not in line, line
        assert line["n_gpus"] == 2 and line["scaling"] == scaling and "RCCL" in line["config"]["sharding"]
        assert line["value"] > 0 and line["roofline"]["frac"] > 0
