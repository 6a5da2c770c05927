#!/usr/bin/env python3
"""Generates tests/golden/cli_fixtures.json from the REFERENCE's own CLI config layer (the two modules
`aggfly/cli/config.py` and `aggfly/cli/preprocess.py` need only the standard library, numpy and yaml, so they load standalone
in the build container although the package as a whole does not import).  Build container only: /root/reference does not
exist on the GPU box, and nothing but this script reads it.  The JSON holds DATA — configs in, the reference's normalised
`RunConfig` fields / `to_aggregator_dict()` / `describe()` text / full error lists out; preprocess expressions in, their values
on a small array or their refusal out — never any of the reference's source text.

    python tests/golden/make_cli_fixtures.py          # rewrites tests/golden/cli_fixtures.json

`tests/test_cli_reference_fixtures.py` holds `aggfly_amd/cli/config.py` and `preprocess.py` to it (N1 of SURVEY.md §8f:
`cli/config.py:98,214-386`, `cli/preprocess.py:143`).
"""
import copy
import dataclasses
import importlib.util
import json
import os

import numpy as np
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def jsonable(x):
    if isinstance(x, np.ndarray):
        return {"__ndarray__": x.tolist(), "dtype": str(x.dtype)}
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    return x


BASE = {
    "regions": {"path": "r.shp", "regionid": "fips"},
    "dataset": {"path": "d_{year}.zarr", "var": "t2m"},
    "aggregate": {"variables": {"tavg": [["aggregate", {"calc": "mean", "groupby": "date"}],
                                         ["aggregate", {"calc": "sum", "groupby": "year"}]]}},
    "years": "2000:2002",
    "output": {"path": "out.parquet"},
}


def cfg(**over):
    """BASE with top-level sections replaced (a value of the string "__drop__" removes the key)."""
    c = copy.deepcopy(BASE)
    for k, v in over.items():
        if v == "__drop__":
            c.pop(k, None)
        else:
            c[k] = v
    return c


def variables(steps, name="v"):
    return {"variables": {name: steps}}


def config_cases():
    agg = lambda **p: ["aggregate", p]
    tf = lambda **p: ["transform", p]
    cases = {}
    for ex in ("era5_counties_area.yaml", "era5_counties_pop.yaml"):       # the shapes of both example configs (BASELINE configs[0] / [2])
        with open(os.path.join(REF, "examples", ex)) as f:
            cases["example:" + ex] = yaml.safe_load(f)
    cases.update({
        "base": cfg(),
        "years_int": cfg(years=1999),
        "years_list": cfg(years=[2001, 2003, "2005"]),
        "years_str_single": cfg(years="1987"),
        "years_range": cfg(years="1980:1983"),
        "years_bad_str": cfg(years="80s"),
        "years_bad_list": cfg(years=[2001, "x"]),
        "years_bool": cfg(years=True),
        "years_float": cfg(years=1999.5),
        "templated_without_years": cfg(years="__drop__"),
        "untemplated_without_years": cfg(dataset={"path": "d.zarr", "var": "t2m"}, years="__drop__"),
        "untemplated_with_years": cfg(dataset={"path": "d.zarr", "var": "t2m"}),
        "dataset_options": cfg(dataset={"path": "d_{year}.nc", "var": "tas", "lon_is_360": False, "timecoord": "t", "xycoords": ["lon", "lat"],
                                        "time_sel": "2000-06", "chunks": {"time": 24}, "clip_to_regions": False, "engine": "zarr",
                                        "storage_options": {"token": "anon"}, "preprocess": "x - 273.15"}),
        "regions_list": cfg(regions={"path": "r.shp", "regionid": "geoid", "region_list": ["01001", "01003"]}),
        "preprocess_from": cfg(dataset={"path": "d_{year}.zarr", "var": "t2m", "preprocess_from": "prep.py:clean"}),
        "preprocess_both": cfg(dataset={"path": "d_{year}.zarr", "var": "t2m", "preprocess": "x - 1", "preprocess_from": "prep.py:clean"}),
        "preprocess_from_no_colon": cfg(dataset={"path": "d_{year}.zarr", "var": "t2m", "preprocess_from": "prep.py"}),
        "xycoords_bad": cfg(dataset={"path": "d_{year}.zarr", "var": "t2m", "xycoords": ["lon"]}),
        "storage_options_bad": cfg(dataset={"path": "d_{year}.zarr", "var": "t2m", "storage_options": "anon"}),
        "reader_engine_bad": cfg(dataset={"path": "d_{year}.zarr", "var": "t2m", "engine": 3}),
        "weights_full": cfg(weights={"project_dir": "./proj", "zero_weight": "drop",
                                     "secondary": {"type": "crop", "path": "crop.tif", "crop": "maize", "feed": "irrigated"}}),
        "weights_zero_weight_bad": cfg(weights={"zero_weight": "zero"}),
        "weights_secondary_not_mapping": cfg(weights={"secondary": "pop.tif"}),
        "weights_secondary_bad": cfg(weights={"secondary": {"type": "people"}}),
        "engine_numba": cfg(aggregate=dict(BASE["aggregate"], engine="numba")),
        "engine_bad": cfg(aggregate=dict(BASE["aggregate"], engine="cuda")),
        "backend_processes": cfg(execution={"backend": "processes", "n_workers": 16, "threads_per_worker": "2"}),
        "backend_bad": cfg(execution={"backend": "mpi"}),
        "output_csv": cfg(output={"path": "o/panel.csv"}),
        "output_pq": cfg(output={"path": "panel.pq"}),
        "output_feather_upper": cfg(output={"path": "PANEL.FEATHER"}),
        "output_format_wins": cfg(output={"path": "panel.dat", "format": "csv"}),
        "output_unknown_ext": cfg(output={"path": "panel.txt"}),
        "output_missing": cfg(output={}),
        "missing_everything": {"aggregate": {}},
        "section_not_mapping": cfg(regions=["r.shp"], execution="threads"),
        "raw_none": None,
        "raw_list": ["regions"],
        "variables_missing": cfg(aggregate={"engine": "auto"}),
        "variables_empty": cfg(aggregate={"variables": {}}),
        "steps_not_list": cfg(aggregate=variables("mean")),
        "steps_empty": cfg(aggregate=variables([])),
        "step_wrong_arity": cfg(aggregate=variables([["aggregate"], agg(calc="sum", groupby="year")])),
        "step_unknown_type": cfg(aggregate=variables([["reduce", {"calc": "mean"}]])),
        "step_params_not_mapping": cfg(aggregate=variables([["aggregate", "mean"]])),
        "agg_bad_calc_and_groupby": cfg(aggregate=variables([agg(calc="median", groupby="season")])),
        "agg_dd_without_ddargs": cfg(aggregate=variables([agg(calc="dd", groupby="date"), agg(calc="sum", groupby="year")])),
        "agg_dd_empty_ddargs": cfg(aggregate=variables([agg(calc="bins", groupby="year", ddargs=[])])),
        "agg_dd": cfg(aggregate=variables([agg(calc="dd", groupby="date", ddargs=[10, 30, 0]), agg(calc="sum", groupby="year")])),
        "agg_multi_dd": cfg(aggregate=variables([agg(calc="mean", groupby="date"),
                                                 agg(calc="bins", groupby="year", ddargs=[[0, 5, 0], [5, 10, 0], [10, 15, 0]])])),
        "agg_sine_week": cfg(aggregate=variables([agg(calc="sine_dd", groupby="week", ddargs=[10, 30, 0])])),
        "tf_exp_list": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(transform="power", exp=[1, 2, 3, 4]), agg(calc="sum", groupby="month")])),
        "tf_exp_int": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(transform="power", exp=2), agg(calc="sum", groupby="year")])),
        "tf_exp_bad": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(transform="power", exp="2"), agg(calc="sum", groupby="year")])),
        "tf_spline": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(transform="spline"), agg(calc="sum", groupby="year")])),
        "tf_spline_key": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(spline=[20]), agg(calc="sum", groupby="year")])),
        "tf_inter": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(inter="other.zarr"), agg(calc="sum", groupby="year")])),
        "tf_nothing": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(transform="log")])),
        "multi_dd_conflict": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(transform="power", exp=[1, 2]),
                                                      agg(calc="bins", groupby="year", ddargs=[[0, 5, 0], [5, 10, 0]])])),
        "multi_dd_after_single_exp": cfg(aggregate=variables([agg(calc="mean", groupby="date"), tf(transform="power", exp=[2]),
                                                              agg(calc="bins", groupby="year", ddargs=[[0, 5, 0], [5, 10, 0]])])),
        "two_variables": cfg(aggregate={"engine": "dask", "variables": {
            "dday": [agg(calc="dd", groupby="date", ddargs=[10, 30, 0]), agg(calc="sum", groupby="year")],
            "tavg": [agg(calc="mean", groupby="date"), tf(transform="power", exp=[1, 2]), agg(calc="sum", groupby="year")]}}),
        "many_errors": {"regions": {"regionid": "fips"}, "dataset": {"var": "t2m", "preprocess": "x", "preprocess_from": "f.py"},
                        "weights": {"zero_weight": "?", "secondary": {"type": "pop"}},
                        "aggregate": {"engine": "gpu", "variables": {"a": [agg(calc="avg", groupby="date")], "b": []}},
                        "years": "x:y", "execution": {"backend": "spark"}, "output": {"path": "o.xlsx"}},
    })
    return cases


PREPROCESS_EXPRS = [
    # builtins (cli/preprocess.py:24-30) and the reference's own test expressions (tests/test_cli.py:280-310) ...
    "identity", "kelvin_to_celsius", "celsius_to_kelvin", "pa_to_kpa", "m_to_mm",
    "x - 273.15", "(x - 32) * 5 / 9", "x ** 2", "-x", "x * 0.1 + 5", "+x", "x % 7", "x // 2", "2 ** x / 1e3", "x - -1", "((x))",
    # ... and its refusals (tests/test_cli.py:312-325) plus neighbours of each rule
    "__import__('os').system('echo hi')", "x.values", "x[0]", "y + 1", "os", "1 + 2", "x if x else 1", "x and 1", "x < 3", "~x", "x @ x",
    "x + 'a'", "x + True", "x + None", "lambda: x", "[x]", "x; x", "", "x +", "kelvin", "abs(x)", "x | 1", "x << 1",
]
PREPROCESS_INPUT = [273.15, 283.15, 0.0, -5.5, 40.0]


def main():
    cfgmod = load("ref_cli_config", os.path.join(REF, "aggfly", "cli", "config.py"))
    ppmod = load("ref_cli_preprocess", os.path.join(REF, "aggfly", "cli", "preprocess.py"))
    out = {"source": "generated by tests/golden/make_cli_fixtures.py from /root/reference/aggfly/cli/config.py and preprocess.py (aggfly v0.2.0)",
           "config": {}, "preprocess": {"input": PREPROCESS_INPUT, "cases": {}}}
    for name, raw in config_cases().items():
        ent = {"raw": raw}
        try:
            rc = cfgmod.parse_config(copy.deepcopy(raw))
        except cfgmod.ConfigError as e:
            ent.update(ok=False, errors=list(e.errors))
        except Exception as e:      # the reference itself falls over (e.g. its multi-dd guard unpacks a malformed step): recorded as such
            ent.update(ok=False, errors=None, crash=f"{type(e).__name__}: {e}")
        else:
            ent.update(ok=True, fields=jsonable(dataclasses.asdict(rc)), templated=rc.templated, resolved_paths=rc.resolved_paths(),
                       aggregator=jsonable(rc.to_aggregator_dict()), describe=cfgmod.describe(rc), check_paths=cfgmod.check_paths(rc))
        out["config"][name] = ent
    x = np.array(PREPROCESS_INPUT)
    for expr in PREPROCESS_EXPRS:
        try:
            f = ppmod.resolve(expr)
            with np.errstate(all="ignore"):
                val = f(x.copy())
        except ppmod.PreprocessError as e:
            out["preprocess"]["cases"][expr] = {"ok": False, "error": str(e)}
        except Exception as e:      # a run-time failure of an expression the reference ACCEPTED (e.g. a string operand)
            out["preprocess"]["cases"][expr] = {"ok": False, "error": None, "crash": f"{type(e).__name__}: {e}"}
        else:
            out["preprocess"]["cases"][expr] = {"ok": True, "value": np.asarray(val, dtype=float).tolist()}
    # resolve()'s other branches
    misc = {}
    misc["none"] = ppmod.resolve(None, None) is None
    for label, args in (("both", ("x - 1", "prep.py:f")), ("not_a_string", (3, None)), ("from_no_colon", (None, "prep.py")),
                        ("from_missing_file", (None, "/no/such/file.py:clean"))):
        try:
            ppmod.resolve(*args)
            misc[label] = {"ok": True}
        except ppmod.PreprocessError as e:
            misc[label] = {"ok": False, "error": str(e)}
    out["preprocess"]["misc"] = misc
    path = os.path.join(HERE, "cli_fixtures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)           # key order is data: the variables of a config keep their order
    n_ok = sum(1 for e in out["config"].values() if e["ok"])
    print(f"{path}: {len(out['config'])} configs ({n_ok} valid), {len(out['preprocess']['cases'])} preprocess expressions")


if __name__ == "__main__":
    main()
