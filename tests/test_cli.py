"""CLI layer: config parsing/validation and the safe preprocess evaluator (CPU), the
CLI == hand-written-API parity of `aggfly/tests/test_cli.py:426-458` (GPU), and the
world_size-2 gloo gather of per-year panels."""
import os
import socket
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import yaml
from click.testing import CliRunner

import aggfly_amd as af
from aggfly_amd.cli import config as cfg, pipeline, preprocess as ppmod
from aggfly_amd.cli.main import cli


def _minimal(**over):
    raw = {
        "regions": {"path": "r.csv", "regionid": "geoid"},
        "dataset": {"path": "d.zarr", "var": "t2m"},
        "aggregate": {"variables": {"tavg": [["aggregate", {"calc": "mean", "groupby": "date"}],
                                              ["transform", {"transform": "power", "exp": [1, 2]}],
                                              ["aggregate", {"calc": "sum", "groupby": "year"}]]}},
        "output": {"path": "out.parquet"},
    }
    raw.update(over)
    return raw


def test_parse_defaults_and_exp_normalisation():
    c = cfg.parse_config(_minimal())
    assert c.engine == "auto" and c.lon_is_360 and c.zero_weight == "nan" and c.output_format == "parquet"
    ad = c.to_aggregator_dict()
    exp = ad["tavg"][1][1]["exp"]
    assert isinstance(exp, np.ndarray) and exp.tolist() == [1, 2]
    assert cfg.parse_config(_minimal(aggregate={"engine": "hip", "variables": _minimal()["aggregate"]["variables"]})).engine == "hip"


def test_years_templating():
    raw = _minimal(dataset={"path": "era5_{year}.zarr", "var": "t2m"}, years="1990:1992")
    c = cfg.parse_config(raw)
    assert c.templated and c.resolved_paths() == ["era5_1990.zarr", "era5_1991.zarr", "era5_1992.zarr"]
    with pytest.raises(cfg.ConfigError, match="no 'years' were given"):
        cfg.parse_config(_minimal(dataset={"path": "era5_{year}.zarr", "var": "t2m"}))


def test_all_errors_are_collected():
    raw = {"regions": {}, "dataset": {"path": "x"}, "output": {"path": "o.xyz"},
           "aggregate": {"engine": "gpu", "variables": {"a": [["aggregate", {"calc": "median", "groupby": "decade"}],
                                                               ["transform", {"transform": "log"}]],
                                                        "b": [["aggregate", {"calc": "dd", "groupby": "date"}]],
                                                        "c": [["aggregate", {"calc": "mean", "groupby": "date"}],
                                                              ["transform", {"transform": "power", "exp": [1, 2]}],
                                                              ["aggregate", {"calc": "bins", "groupby": "year", "ddargs": [[0, 1, 0], [1, 2, 0]]}]]}}}
    with pytest.raises(cfg.ConfigError) as e:
        cfg.parse_config(raw)
    msg = str(e.value)
    for frag in ("regions.path is required", "regions.regionid is required", "dataset.var is required", "aggregate.engine",
                 "calc 'median'", "groupby 'decade'", "transform step needs", "requires a non-empty 'ddargs'",
                 "cannot combine a multi-'ddargs'", "output.format"):
        assert frag in msg, frag
    assert len(e.value.errors) >= 9


def test_preprocess_builtins_and_expressions():
    assert np.allclose(ppmod.resolve("kelvin_to_celsius")(np.array([273.15, 283.15])), [0.0, 10.0])
    for expr, inp, want in [("x - 273.15", [273.15, 283.15], [0.0, 10.0]), ("(x - 32) * 5 / 9", [32.0, 212.0], [0.0, 100.0]),
                            ("x ** 2", [2.0, 3.0], [4.0, 9.0]), ("-x", [1.0, -2.0], [-1.0, 2.0])]:
        assert np.allclose(ppmod.resolve(expr)(np.array(inp)), want)
    for bad in ("__import__('os').system('echo hi')", "x.values", "x[0]", "y + 1", "os", "1 + 2"):
        with pytest.raises(ppmod.PreprocessError):
            ppmod.resolve(bad)
    assert ppmod.resolve(None, None) is None
    with pytest.raises(ppmod.PreprocessError):
        ppmod.resolve("x - 1", "prep.py:f")


def test_preprocess_from_file(tmp_path):
    mod = tmp_path / "prep.py"
    mod.write_text("def clean(x):\n    return x - 273.15\n")
    assert np.allclose(ppmod.resolve(None, f"{mod}:clean")(np.array([273.15, 300.15])), [0.0, 27.0])
    with pytest.raises(ppmod.PreprocessError):
        ppmod.resolve(None, f"{mod}:nope")


def test_validate_command(tmp_path):
    good = tmp_path / "good.yaml"
    good.write_text(yaml.safe_dump(_minimal()))
    r = CliRunner().invoke(cli, ["validate", str(good)])
    assert r.exit_code == 0 and "OK" in r.output
    bad = tmp_path / "bad.yaml"
    bad.write_text(yaml.safe_dump({"regions": {}}))
    r = CliRunner().invoke(cli, ["validate", str(bad)])
    assert r.exit_code == 1 and "Config is invalid:" in r.output
    # the reference's report (`cli/main.py:89-125`): normalised plan, unresolved local paths as warnings, --strict -> errors
    r = CliRunner().invoke(cli, ["validate", str(good)])
    assert "Normalized plan" in r.output and "Config OK." in r.output and "area-only" in r.output
    assert "Warnings:" in r.output and "regions.path does not exist" in r.output and "dataset.path does not resolve" in r.output
    r = CliRunner().invoke(cli, ["validate", str(good), "--strict"])
    assert r.exit_code == 1 and "Errors:" in r.output and "Config OK." not in r.output
    for name, steps in _minimal()["aggregate"]["variables"].items():
        assert f"- {name}: " in CliRunner().invoke(cli, ["validate", str(good)]).output
    # every flag of the reference's `run` exists (values checked by click before anything is read)
    r = CliRunner().invoke(cli, ["run", "--help"])
    for flag in ("-o, --output", "--engine", "--years", "--project-dir", "--backend", "--n-workers", "-v, --verbose"):
        assert flag in r.output, flag
    r = CliRunner().invoke(cli, ["weights", "--help"])
    assert "--project-dir" in r.output and "--verbose" in r.output
    r = CliRunner().invoke(cli, ["regions", "--help"])
    assert "--rows" in r.output and "--uniqueness" in r.output


def _write_run_inputs(tmp_path, years=(2000,)):
    """8x8 grid, seed 7, a box region well inside it (test_cli.py:375-404); the weights table is
    what aggfly's area weights give for that box: whole cells 1, edge cells their overlap."""
    lon = np.arange(-100.0, -60.0, 5.0)
    lat = np.arange(20.0, 60.0, 5.0)
    np.random.seed(7)
    paths = []
    for y in years:
        time = pd.date_range(f"{y}-07-01", periods=4, freq="12h")
        arr = np.random.normal(20, 15, (len(time), len(lat), len(lon))) + 273.15
        ds = af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}),
                        lon_is_360=False)
        p = str(tmp_path / f"ds_{y}.zarr")
        af.dataset_to_zarr(ds, p, var="t2m")
        paths.append(p)
    box = (-92.0, 28.0, -68.0, 52.0)
    regions = pd.DataFrame({"geoid": ["r1"], "minx": [box[0]], "miny": [box[1]], "maxx": [box[2]], "maxy": [box[3]]})
    rpath = str(tmp_path / "regions.csv")
    regions.to_csv(rpath, index=False)
    # clipped grid = cells whose centroid lies within half a cell of the box (grid.py:176-217)
    keep_lon = lon[(lon >= box[0] - 2.5) & (lon <= box[2] + 2.5)]
    keep_lat = lat[(lat >= box[1] - 2.5) & (lat <= box[3] + 2.5)]
    rows = []
    for iy, la in enumerate(keep_lat):
        for ix, lo in enumerate(keep_lon):
            ox = max(0.0, min(lo + 2.5, box[2]) - max(lo - 2.5, box[0])) / 5.0
            oy = max(0.0, min(la + 2.5, box[3]) - max(la - 2.5, box[1])) / 5.0
            if ox * oy > 0:
                rows.append((iy * len(keep_lon) + ix, 0, ox * oy * np.cos(np.deg2rad(la))))
    tab = pd.DataFrame(rows, columns=["cell_id", "index_right", "weight"])
    wpath = str(tmp_path / "weights.parquet")
    tab.to_parquet(wpath, index=False)
    return paths, rpath, wpath


def _run_config(dpath, rpath, wpath, out, **extra):
    c = {"regions": {"path": rpath, "regionid": "geoid"},
         "dataset": {"path": dpath, "var": "t2m", "lon_is_360": False, "preprocess": "kelvin_to_celsius"},
         "weights": {"table": wpath},
         "aggregate": {"engine": "auto", "variables": {"tavg": [["aggregate", {"calc": "mean", "groupby": "date"}],
                                                                ["transform", {"transform": "power", "exp": [1, 2]}],
                                                                ["aggregate", {"calc": "sum", "groupby": "month"}]]}},
         "output": {"path": out}}
    c.update(extra)
    return c


@pytest.mark.gpu
def test_run_matches_direct_api(torch_cuda, tmp_path):
    paths, rpath, wpath = _write_run_inputs(tmp_path, years=(2000, 2001))
    out = str(tmp_path / "panel.parquet")
    cpath = tmp_path / "config.yaml"
    cpath.write_text(yaml.safe_dump(_run_config(str(tmp_path / "ds_{year}.zarr"), rpath, wpath, out, years="2000:2001")))
    result = CliRunner().invoke(cli, ["run", str(cpath)])
    assert result.exit_code == 0, result.output
    actual = pd.read_parquet(out)
    # the equivalent hand-written af.* script
    gr = af.weights.georegions_from_path(rpath, "geoid")
    frames = []
    for p in paths:
        ds = af.dataset_from_path(p, var="t2m", lon_is_360=False, georegions=gr, name="t2m", preprocess=lambda x: x - 273.15)
        w = af.weights_from_objects(ds, gr, table=pd.read_parquet(wpath))
        w.calculate_weights()
        frames.append(af.aggregate_dataset(dataset=ds, weights=w,
                                           tavg=[("aggregate", {"calc": "mean", "groupby": "date"}),
                                                 ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                                                 ("aggregate", {"calc": "sum", "groupby": "month"})]))
    expected = pd.concat(frames, ignore_index=True)
    assert list(actual.columns) == list(expected.columns) and len(actual) == 2
    assert np.allclose(actual[["tavg_1", "tavg_2"]].values, expected[["tavg_1", "tavg_2"]].values)
    # clip on/off must not change results when the weights table addresses the unclipped grid... the
    # table here addresses the CLIPPED grid, so only check the clipped run against the oracle
    from oracle import ref_aggregate as ra
    ds = af.dataset_from_path(paths[0], var="t2m", lon_is_360=False, georegions=gr, preprocess=lambda x: x - 273.15)
    ow = ra.OWeights(pd.read_parquet(wpath), np.arange(len(ds.latitude) * len(ds.longitude)), gr.shp["geoid"], "geoid", "nan")
    want = ra.aggregate_dataset(ow, ra.ODataset(ds.cube(), ds.time, ds.latitude, ds.longitude, False), engine="numba",
                                **cfg.parse_config(yaml.safe_load(cpath.read_text())).to_aggregator_dict())
    np.testing.assert_allclose(actual[["tavg_1", "tavg_2"]].values[:1], want[["tavg_1", "tavg_2"]].values, rtol=1e-12)


@pytest.mark.gpu
def test_run_single_store_is_cut_inside_the_store(torch_cuda, tmp_path):
    """A config with ONE un-templated store takes the single-store route: ranks (and HBM-budget windows) take runs of
    output periods of the same store.  One process, one process under a tiny HBM budget, and two ranks all write the
    frame the hand-written API call gives."""
    _, rpath, wpath = _write_run_inputs(tmp_path)
    lon, lat = np.arange(-100.0, -60.0, 5.0), np.arange(20.0, 60.0, 5.0)
    time = pd.date_range("2001-11-15", periods=300, freq="12h")                     # Nov 2001 .. Apr 2002: 6 monthly periods
    arr = np.random.default_rng(11).normal(20, 15, (len(time), len(lat), len(lon))) + 273.15
    arr[37, 3, 4] = np.nan
    store = str(tmp_path / "long.zarr")
    af.dataset_to_zarr(af.Dataset(af.DataArray(arr, ["time", "latitude", "longitude"], {"time": time, "latitude": lat, "longitude": lon}),
                                  lon_is_360=False), store, var="t2m", chunks={"time": 40, "latitude": 8, "longitude": 8})
    gr = af.weights.georegions_from_path(rpath, "geoid")
    ds = af.dataset_from_path(store, var="t2m", lon_is_360=False, georegions=gr, name="t2m", preprocess=lambda x: x - 273.15)
    w = af.weights_from_objects(ds, gr, table=pd.read_parquet(wpath))
    expected = af.aggregate_dataset(dataset=ds, weights=w, tavg=[("aggregate", {"calc": "mean", "groupby": "date"}),
                                                                 ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                                                                 ("aggregate", {"calc": "sum", "groupby": "month"})])
    assert len(expected) == 6                                                        # the NaN cell only leaves its month's average
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = [sys.executable, "-m", "aggfly_amd.cli.main"]
    two = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "aggfly_amd.cli.main"]
    step = 8 * 8 * 8                                                                 # bytes of one stored time step
    for tag, cmd, extra in (("one", one, {}), ("windows", one, {"AGGFLY_HIP_WINDOW_BYTES": str(70 * step)}),
                            ("ranks", two, {"AGGFLY_DIST_BACKEND": "gloo"})):
        out = str(tmp_path / f"panel_{tag}.csv")
        cpath = tmp_path / f"config_{tag}.yaml"
        cpath.write_text(yaml.safe_dump(_run_config(store, rpath, wpath, out)))
        r = subprocess.run(cmd + ["run", str(cpath), "-v"], capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=root, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "streamed through HBM in windows" in r.stdout, r.stdout
        got = pd.read_csv(out, parse_dates=["time"])
        assert list(got.columns) == list(expected.columns) and got["geoid"].tolist() == expected["geoid"].tolist(), tag
        assert got["time"].tolist() == pd.DatetimeIndex(expected["time"]).tolist(), tag
        np.testing.assert_allclose(got[["tavg_1", "tavg_2"]].values, expected[["tavg_1", "tavg_2"]].values, rtol=1e-13, err_msg=tag)
    # ONE output period (a single year, annual sums) and two ranks: the cells are sharded instead (latitude bands, one all_reduce)
    short = str(tmp_path / "one_year.zarr")
    sel = slice(100, 300)                                                            # all of it inside 2002
    af.dataset_to_zarr(af.Dataset(af.DataArray(arr[sel], ["time", "latitude", "longitude"], {"time": time[sel], "latitude": lat, "longitude": lon}),
                                  lon_is_360=False), short, var="t2m", chunks={"time": 200, "latitude": 4, "longitude": 4})
    annual = {"engine": "auto", "variables": {"tavg": [["aggregate", {"calc": "mean", "groupby": "date"}],
                                                       ["transform", {"transform": "power", "exp": [1, 2]}],
                                                       ["aggregate", {"calc": "sum", "groupby": "year"}]]}}
    outs = {}
    for tag, cmd, extra in (("y1", one, {}), ("y2", two[:8] + [str(_free_port())] + two[9:], {"AGGFLY_DIST_BACKEND": "gloo"})):
        out = str(tmp_path / f"panel_{tag}.csv")
        cpath = tmp_path / f"config_{tag}.yaml"
        cpath.write_text(yaml.safe_dump(_run_config(short, rpath, wpath, out, aggregate=annual)))
        r = subprocess.run(cmd + ["run", str(cpath), "-v"], capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=root, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("latitude bands" in r.stdout) == (tag == "y2"), r.stdout
        outs[tag] = pd.read_csv(out)
    assert len(outs["y1"]) == 1
    pd.testing.assert_frame_equal(outs["y1"], outs["y2"], rtol=1e-12, atol=0)
    # dataset.time_sel on the single-store routes: the long store restricted to 2002 — monthly periods over two ranks
    # (time route) and one annual period over two ranks (latitude bands) — equals the API call on Dataset(time_sel="2002")
    ds_sel = af.dataset_from_path(store, var="t2m", lon_is_360=False, georegions=gr, name="t2m", preprocess=lambda x: x - 273.15, time_sel="2002")
    for tag, agg_cfg, groupby in (("selm", None, "month"), ("sely", annual, "year")):
        want = af.aggregate_dataset(dataset=ds_sel, weights=w, tavg=[("aggregate", {"calc": "mean", "groupby": "date"}),
                                                                      ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
                                                                      ("aggregate", {"calc": "sum", "groupby": groupby})])
        out = str(tmp_path / f"panel_{tag}.csv")
        cpath = tmp_path / f"config_{tag}.yaml"
        raw = _run_config(store, rpath, wpath, out) if agg_cfg is None else _run_config(store, rpath, wpath, out, aggregate=agg_cfg)
        raw["dataset"]["time_sel"] = "2002"
        cpath.write_text(yaml.safe_dump(raw))
        cmd = two[:8] + [str(_free_port())] + two[9:]
        r = subprocess.run(cmd + ["run", str(cpath), "-v"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PYTHONPATH=root, AGGFLY_DIST_BACKEND="gloo"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("latitude bands" in r.stdout) == (tag == "sely"), r.stdout
        got = pd.read_csv(out, parse_dates=["time"])
        assert len(got) == len(want) and got["time"].tolist() == pd.DatetimeIndex(want["time"]).tolist(), tag
        np.testing.assert_allclose(got[["tavg_1", "tavg_2"]].values, want[["tavg_1", "tavg_2"]].values, rtol=1e-12, err_msg=tag)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _gather_worker(rank, ws, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        def frame(i):
            t = pd.date_range(f"{2000 + i}-01-31", periods=2, freq="ME")
            return pd.DataFrame({"geoid": ["a", "a", "b"][: 3 - (i % 2)], "time": list(t)[: 2] + list(t)[:1][: 1 - (i % 2)],
                                 "v1": np.arange(3 - (i % 2)) + 0.5 + i, "v2": [np.nan, 1.0, 2.0][: 3 - (i % 2)]})
        n = 5
        mine = {i: frame(i) for i in range(n)[rank::ws]}
        full = pipeline._gather_frames(mine, n, "geoid")
        for i in range(n):
            pd.testing.assert_frame_equal(full[i], frame(i))
        q.put((rank, "ok"))
    except Exception as e:
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


def test_year_scheduler_gather_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(out) == [(0, "ok"), (1, "ok")], out


@pytest.mark.gpu
def test_cli_two_ranks_equals_one(torch_cuda, tmp_path):
    """The year loop as a 2-rank job (ranks share the box's GPU, gloo exchange) writes the same
    panel as the single-process run."""
    import subprocess
    import sys
    paths, rpath, wpath = _write_run_inputs(tmp_path, years=(2000, 2001, 2002))
    outs = []
    for tag, cmd in (("one", [sys.executable, "-m", "aggfly_amd.cli.main"]),
                     ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", "aggfly_amd.cli.main"])):
        out = str(tmp_path / f"panel_{tag}.csv")
        cpath = tmp_path / f"config_{tag}.yaml"
        cpath.write_text(yaml.safe_dump(_run_config(str(tmp_path / "ds_{year}.zarr"), rpath, wpath, out, years="2000:2002")))
        env = dict(os.environ, AGGFLY_DIST_BACKEND="gloo", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run(cmd + ["run", str(cpath), "--quiet"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(pd.read_csv(out))
    pd.testing.assert_frame_equal(outs[0], outs[1])
    assert len(outs[0]) == 3


def test_info_regions_weights_commands_need_no_gpu(tmp_path):
    paths, rpath, wpath = _write_run_inputs(tmp_path)
    r = CliRunner().invoke(cli, ["info", paths[0], "--var", "t2m"])
    assert r.exit_code == 0, r.output
    assert "zarr store" in r.output and "t2m" in r.output and "lon_is_360: False" in r.output and "4 steps" in r.output
    r = CliRunner().invoke(cli, ["regions", rpath, "--regionid", "geoid"])
    assert r.exit_code == 0 and "1 regions" in r.output and "r1" in r.output
    cpath = tmp_path / "c.yaml"
    cpath.write_text(yaml.safe_dump(_run_config(paths[0], rpath, wpath, str(tmp_path / "o.csv"))))
    r = CliRunner().invoke(cli, ["weights", str(cpath)])
    assert r.exit_code == 0, r.output
    assert "(cell, region) pairs" in r.output and "1 regions" in r.output
    # a config without any weights source explains where weights come from
    bad = _run_config(paths[0], rpath, wpath, str(tmp_path / "o.csv"))
    bad["weights"] = {}
    cpath.write_text(yaml.safe_dump(bad))
    r = CliRunner().invoke(cli, ["weights", str(cpath)])
    assert r.exit_code == 1 and "no precomputed weights found" in r.output


def test_single_store_route_needs_a_gpu_and_a_plain_store(tmp_path):
    """`pipeline.store_route_ok`: the cut-inside-the-store route is for ONE un-templated local Zarr / netCDF-4 store with
    nestable output frequencies — and only with a GPU; everything else keeps the per-path scheduler."""
    from aggfly_amd import hip
    paths, rpath, wpath = _write_run_inputs(tmp_path)
    c = cfg.parse_config(_run_config(paths[0], rpath, wpath, str(tmp_path / "o.csv")))
    assert pipeline.store_route_ok(c, c.resolved_paths()) == (hip.device_count() > 0)
    assert not pipeline.store_route_ok(c, [paths[0], paths[0]])                       # several paths: the year scheduler
    glob_cfg = cfg.parse_config(_run_config(str(tmp_path / "ds_*.zarr"), rpath, wpath, str(tmp_path / "o.csv")))
    assert not pipeline.store_route_ok(glob_cfg, glob_cfg.resolved_paths())
    weekly = _run_config(paths[0], rpath, wpath, str(tmp_path / "o.csv"))
    weekly["aggregate"] = {"variables": {"t": [["aggregate", {"calc": "mean", "groupby": "week"}]]}}
    wk = cfg.parse_config(weekly)
    assert not pipeline.store_route_ok(wk, wk.resolved_paths())                       # weeks do not nest: no time sharding


def test_weights_table_found_in_project_cache(tmp_path):
    """N3: the reference caches weights as {project_dir}/tmp/GridWeights/mod-<sha>/<sha>.feather
    (`aggfly/cache/project_cache.py:46-47`); the CLI picks that file up when weights.table is unset."""
    paths, rpath, wpath = _write_run_inputs(tmp_path)
    cache = tmp_path / "proj" / "tmp" / "GridWeights" / "mod-abc123"
    cache.mkdir(parents=True)
    pd.read_parquet(wpath).to_feather(cache / "abc123.feather")
    raw = _run_config(paths[0], rpath, wpath, str(tmp_path / "o.csv"))
    raw["weights"] = {"project_dir": str(tmp_path / "proj")}
    c = cfg.parse_config(raw)
    assert pipeline.find_weights_table(c).endswith("abc123.feather")
    w, gr, sample = pipeline.compute_weights(c)
    assert len(w.weights) == len(pd.read_parquet(wpath)) and w.zero_weight == "nan"
    assert len(sample.latitude) < 8 and len(sample.longitude) < 8        # clipped to the regions' extent
