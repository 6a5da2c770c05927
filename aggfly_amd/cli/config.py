"""YAML config -> ``RunConfig`` for the CLI.

Accepts the reference CLI's schema (`aggfly/cli/config.py:214-386`; example
`examples/era5_counties_area.yaml`) so existing configs keep working, plus two extensions:
``aggregate.engine: hip`` and ``weights.table`` (path of the precomputed weights table).

Design: the schema is DATA (``SCHEMA`` below: section -> field -> rule), and one generic walker
validates it, so that every problem in a file is reported in one pass (``ConfigError.errors``)
and adding a field is a one-line change.  Only the step lists under ``aggregate.variables``
need hand-written checks.  Nothing here touches climate data.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field, fields
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import yaml

ALLOWED_CALCS = {"mean", "nanmean", "sum", "min", "max", "dd", "bins", "sine_dd"}
CALCS_NEEDING_DDARGS = {"dd", "bins", "sine_dd"}
ALLOWED_GROUPBY = {"date", "month", "year", "week"}
ALLOWED_ENGINE = {"auto", "dask", "numba", "hip"}
ALLOWED_BACKEND = {"threads", "processes", "none"}
ALLOWED_FORMAT = {"parquet", "feather", "csv"}
ALLOWED_SECONDARY = {"pop", "crop", "generic"}
ALLOWED_ZERO_WEIGHT = {"nan", "area", "drop"}
ALLOWED_STEP_TYPES = {"aggregate", "transform"}


class ConfigError(Exception):
    """Validation failure carrying every message found."""

    def __init__(self, errors):
        self.errors = list(errors)
        super().__init__("\n".join(f"- {e}" for e in self.errors))


@dataclass
class Rule:
    """How one ``section.key`` maps onto a RunConfig attribute."""
    attr: str
    default: Any = None
    required: bool = False
    choices: Optional[set] = None
    kind: Optional[type] = None          # dict / str / list: checked when the value is present
    kind_hint: str = ""
    cast: Optional[type] = None


# section -> yaml key -> rule
SCHEMA: Dict[str, Dict[str, Rule]] = {
    "regions": {
        "path": Rule("regions_path", required=True),
        "regionid": Rule("regionid", required=True),
        "region_list": Rule("region_list"),
    },
    "dataset": {
        "path": Rule("dataset_path", required=True),
        "var": Rule("var", required=True),
        "preprocess": Rule("preprocess"),
        "preprocess_from": Rule("preprocess_from"),
        "lon_is_360": Rule("lon_is_360", default=True, cast=bool),
        "timecoord": Rule("timecoord", default="time"),
        "xycoords": Rule("xycoords", default=["longitude", "latitude"]),
        "time_sel": Rule("time_sel"),
        "chunks": Rule("chunks"),
        "clip_to_regions": Rule("clip_to_regions", default=True, cast=bool),
        "storage_options": Rule("storage_options", kind=dict, kind_hint="a mapping"),
        "engine": Rule("reader_engine", kind=str, kind_hint="a string (e.g. 'zarr')"),
    },
    "weights": {
        "project_dir": Rule("project_dir"),
        "table": Rule("weights_table"),
        "zero_weight": Rule("zero_weight", default="nan", choices=ALLOWED_ZERO_WEIGHT),
        "secondary": Rule("secondary"),
    },
    "aggregate": {
        "engine": Rule("engine", default="auto", choices=ALLOWED_ENGINE),
        "variables": Rule("variables"),
    },
    "execution": {
        "backend": Rule("backend", default="threads", choices=ALLOWED_BACKEND),
        "n_workers": Rule("n_workers", default=1, cast=int),
        "threads_per_worker": Rule("threads_per_worker", default=1, cast=int),
    },
    "output": {
        "path": Rule("output_path", required=True),
        "format": Rule("output_format"),
    },
}


@dataclass
class SecondaryWeightsConfig:
    type: str
    path: str
    crop: Optional[str] = None
    feed: Optional[str] = None


@dataclass
class RunConfig:
    regions_path: str = ""
    regionid: str = ""
    region_list: Optional[List[str]] = None
    dataset_path: str = ""
    var: str = ""
    preprocess: Optional[str] = None
    preprocess_from: Optional[str] = None
    lon_is_360: bool = True
    timecoord: str = "time"
    xycoords: Tuple[str, str] = ("longitude", "latitude")
    time_sel: Optional[str] = None
    chunks: Optional[Dict[str, object]] = None
    clip_to_regions: bool = True
    storage_options: Optional[Dict[str, object]] = None
    reader_engine: Optional[str] = None
    project_dir: Optional[str] = None
    weights_table: Optional[str] = None
    secondary: Optional[SecondaryWeightsConfig] = None
    zero_weight: str = "nan"
    engine: str = "auto"
    variables: Dict[str, List] = field(default_factory=dict)
    years: Optional[List[int]] = None
    backend: str = "threads"
    n_workers: int = 1
    threads_per_worker: int = 1
    output_path: str = ""
    output_format: str = ""

    @property
    def templated(self) -> bool:
        return "{year}" in self.dataset_path

    def resolved_paths(self) -> List[str]:
        if not self.templated:
            return [self.dataset_path]
        return [self.dataset_path.format(year=y) for y in (self.years or [])]

    def to_aggregator_dict(self) -> Dict[str, List]:
        """``variables`` as the ``aggregator_dict`` of aggregate_dataset.  A power transform's
        ``exp`` is handed over as a NumPy array: the library wraps non-lists and then indexes
        ``[0]``, so a plain list [1, 2] would be read as the scalar 1 (`config.py:98-113`)."""
        def norm(kind, params):
            params = dict(params)
            if kind == "transform" and "exp" in params:
                params["exp"] = np.array(params["exp"])
            return kind, params
        return {name: [norm(k, p) for k, p in steps] for name, steps in self.variables.items()}


# --------------------------------------------------------------------------------------
# field-level helpers
# --------------------------------------------------------------------------------------
def _parse_years(spec, errors):
    """``"1980:1990"`` (inclusive) | list | int | None -> list of ints; the reference's message per kind of mistake
    (`cli/config.py:116-143`)."""
    if spec is None:
        return None
    problem = None
    if isinstance(spec, bool):                      # YAML `true` is an int subclass
        problem = "must be a range 'start:end', a list, or an int"
    elif isinstance(spec, int):
        return [spec]
    elif isinstance(spec, list):
        try:
            return [int(y) for y in spec]
        except (TypeError, ValueError):
            problem = f"list must contain integers, got {spec!r}"
    elif isinstance(spec, str):
        lo, sep, hi = spec.partition(":")
        try:
            return list(range(int(lo), int(hi) + 1)) if sep else [int(lo)]
        except ValueError:
            problem = f"could not parse {spec!r} (use 'start:end' or an int)"
    else:
        problem = f"unsupported type {type(spec).__name__}"
    errors.append(f"years: {problem}")
    return None


def _check_steps(name, steps, errors):
    where = f"aggregate.variables.{name}"
    if not isinstance(steps, list) or not steps:
        errors.append(f"{where}: must be a non-empty list of steps")
        return
    fan_out = 1
    for i, step in enumerate(steps):
        loc = f"{where}[{i}]"
        if not (isinstance(step, (list, tuple)) and len(step) == 2):
            errors.append(f"{loc}: each step must be [step_type, params]")
            continue
        kind, params = step
        if kind not in ALLOWED_STEP_TYPES:
            errors.append(f"{loc}: unknown step type {kind!r} (expected one of {sorted(ALLOWED_STEP_TYPES)})")
        elif not isinstance(params, dict):
            errors.append(f"{loc}: params must be a mapping")
        elif kind == "aggregate":
            calc, dd = params.get("calc"), params.get("ddargs")
            if calc not in ALLOWED_CALCS:
                errors.append(f"{loc}: calc {calc!r} not in {sorted(ALLOWED_CALCS)}")
            if params.get("groupby") not in ALLOWED_GROUPBY:
                errors.append(f"{loc}: groupby {params.get('groupby')!r} not in {sorted(ALLOWED_GROUPBY)}")
            if calc in CALCS_NEEDING_DDARGS:
                if not isinstance(dd, list) or not dd:
                    errors.append(f"{loc}: calc {calc!r} requires a non-empty 'ddargs' list")
                elif isinstance(dd[0], list) and fan_out > 1:
                    errors.append(f"{where}: cannot combine a multi-'ddargs' (bins) step with a multi-output "
                                  "transform (e.g. multiple exponents) — the library rejects this at runtime")
        else:
            spline = params.get("transform") == "spline" or "spline" in params
            if not ("exp" in params or "inter" in params or spline):
                errors.append(f"{loc}: transform step needs one of 'exp' (power), 'inter', or transform: spline")
            exp = params.get("exp")
            if exp is not None and not isinstance(exp, (list, int)):
                errors.append(f"{loc}: 'exp' must be an int or a list of ints")
            if isinstance(exp, list):
                fan_out = len(exp)


def _secondary(raw, errors):
    if raw is None:
        return None
    if not isinstance(raw, dict):
        errors.append("weights.secondary must be a mapping")
        return None
    if raw.get("type") not in ALLOWED_SECONDARY:
        errors.append(f"weights.secondary.type {raw.get('type')!r} not in {sorted(ALLOWED_SECONDARY)}")
    if not raw.get("path"):
        errors.append("weights.secondary.path is required")
    return SecondaryWeightsConfig(raw.get("type"), raw.get("path"), raw.get("crop"), raw.get("feed"))


# --------------------------------------------------------------------------------------
# the walker
# --------------------------------------------------------------------------------------
def _after_preprocess(values, raw, errors):
    if values["preprocess"] is not None and values["preprocess_from"] is not None:
        errors.append("dataset: set at most one of 'preprocess' and 'preprocess_from'")
    if values["preprocess_from"] is not None and ":" not in str(values["preprocess_from"]):
        errors.append("dataset.preprocess_from must be 'path/to/file.py:function'")


def _after_xycoords(values, raw, errors):
    xy = values["xycoords"]
    if not (isinstance(xy, list) and len(xy) == 2):
        errors.append("dataset.xycoords must be a 2-item list [lon_name, lat_name]")
        xy = ["longitude", "latitude"]
    values["xycoords"] = (xy[0], xy[1])


def _after_secondary(values, raw, errors):
    values["secondary"] = _secondary(values["secondary"], errors)


def _after_variables(values, raw, errors):
    variables = values["variables"]
    if not isinstance(variables, dict) or not variables:
        errors.append("aggregate.variables must be a non-empty mapping of name -> steps")
        values["variables"] = {}
    else:
        for name, steps in variables.items():
            _check_steps(name, steps, errors)
    values["years"] = _parse_years(raw.get("years"), errors)      # job control follows the pipeline stages, as in the reference's report


def _after_output(values, raw, errors):
    out_path, fmt = values["output_path"], values["output_format"]
    if fmt is None and out_path:
        ext = os.path.splitext(str(out_path))[1].lstrip(".").lower()
        fmt = {"pq": "parquet"}.get(ext, ext)
    if fmt not in ALLOWED_FORMAT:
        errors.append(f"output.format {fmt!r} not in {sorted(ALLOWED_FORMAT)} "
                      "(set output.format or use a .parquet/.feather/.csv extension)")
    values["output_format"] = fmt
    if values["dataset_path"] and "{year}" in str(values["dataset_path"]) and not values["years"]:
        errors.append("dataset.path contains '{year}' but no 'years' were given (add years: 'start:end')")


# cross-field rules, run right after the field they belong to so that the messages come out in the order of the file's
# sections — the order the reference reports them in (`cli/config.py:214-386`; pinned by tests/golden/cli_fixtures.json)
AFTER = {("dataset", "preprocess_from"): _after_preprocess, ("dataset", "xycoords"): _after_xycoords,
         ("weights", "secondary"): _after_secondary, ("aggregate", "variables"): _after_variables, ("output", "format"): _after_output}


def parse_config(raw) -> RunConfig:
    """Validate a parsed YAML mapping; raise ConfigError listing EVERY problem found."""
    if not isinstance(raw, dict) or not raw:
        raise ConfigError(["config must be a non-empty YAML mapping"])
    errors: List[str] = []
    values: Dict[str, Any] = {}
    bodies = {}
    for section in SCHEMA:                              # malformed sections first, then the fields section by section
        body = raw.get(section)
        if body is None:
            body = {}
        if not isinstance(body, dict):
            errors.append(f"{section}: must be a mapping")
            body = {}
        bodies[section] = body
    for section, rules in SCHEMA.items():
        body = bodies[section]
        for key, rule in rules.items():
            val = body.get(key, rule.default)
            if rule.required and not val:
                errors.append(f"{section}.{key} is required")
            if val is not None and rule.kind is not None and not isinstance(val, rule.kind):
                errors.append(f"{section}.{key} must be {rule.kind_hint}")
                val = None
            if val is not None and rule.choices is not None and val not in rule.choices:
                errors.append(f"{section}.{key} {val!r} not in {sorted(rule.choices)}")
                val = rule.default
            if val is not None and rule.cast is not None:
                val = rule.cast(val)
            values[rule.attr] = val
            hook = AFTER.get((section, key))
            if hook is not None:
                hook(values, raw, errors)
    if errors:
        raise ConfigError(errors)
    known = {f.name for f in fields(RunConfig)}
    return RunConfig(**{k: v for k, v in values.items() if k in known})


def load_config(path) -> RunConfig:
    try:
        with open(path) as f:
            raw = yaml.safe_load(f)
    except FileNotFoundError:
        raise ConfigError([f"config file not found: {path}"])
    except yaml.YAMLError as e:
        raise ConfigError([f"could not parse YAML: {e}"])
    return parse_config(raw)


# --------------------------------------------------------------------------------------
# what `aggfly validate` prints (`cli/config.py:406-466` in the reference)
# --------------------------------------------------------------------------------------
def _is_remote(path) -> bool:
    return isinstance(path, str) and "://" in path


def check_paths(config: RunConfig) -> List[str]:
    """Messages for local inputs that do not exist (remote URLs are never fetched)."""
    import glob
    import os
    missing = []
    if not _is_remote(config.regions_path) and not os.path.exists(config.regions_path):
        missing.append(f"regions.path does not exist: {config.regions_path}")
    missing += [f"dataset.path does not resolve: {p}" for p in config.resolved_paths()
                if not _is_remote(p) and not glob.glob(p) and not os.path.exists(p)]
    sec = config.secondary
    if sec is not None and not _is_remote(sec.path) and not os.path.exists(sec.path):
        missing.append(f"weights.secondary.path does not exist: {sec.path}")
    if config.weights_table and not _is_remote(config.weights_table) and not os.path.exists(config.weights_table):
        missing.append(f"weights.table does not exist: {config.weights_table}")
    return missing


def describe(config: RunConfig) -> str:
    """The normalised plan of a config, one fact per line."""
    def step(kind, params):
        what = params.get("calc") or params.get("transform") or "?"
        return f"{kind}:{what}" + (f"@{params['groupby']}" if params.get("groupby") else "")

    rows = [("regions", f"{config.regions_path}  (id column: {config.regionid})"),
            ("dataset", f"{config.dataset_path}  var={config.var}"),
            ("", f"lon_is_360={config.lon_is_360} timecoord={config.timecoord} xycoords={list(config.xycoords)}")]
    if config.reader_engine:
        rows.append(("", f"reader engine: {config.reader_engine}"))
    if config.storage_options:      # keys only: the values carry credentials
        rows.append(("", "storage_options: {" + ", ".join(sorted(config.storage_options)) + "} (values hidden)"))
    if config.preprocess:
        rows.append(("", f"preprocess: {config.preprocess}"))
    elif config.preprocess_from:
        rows.append(("", f"preprocess_from: {config.preprocess_from}"))
    if config.templated:
        yrs = config.years or []
        rows.append(("years", f"{yrs[0]}..{yrs[-1]} ({len(yrs)} files)" if yrs else "(none)"))
    if config.secondary is not None:
        rows.append(("weights", f"{config.secondary.type} secondary ({config.secondary.path})"))
    else:
        rows.append(("weights", "area-only"))
    if config.weights_table:
        rows.append(("", f"precomputed table: {config.weights_table}"))
    rows += [("zero wt", config.zero_weight),
             ("engine", f"{config.engine}   backend: {config.backend}"),
             ("output", f"{config.output_path}  ({config.output_format})"),
             ("variables", str(len(config.variables)))]
    lines = ["Normalized plan"] + [(f"  {k:<10}: {v}" if k else f"              {v}") for k, v in rows]
    lines += [f"    - {name}: " + " -> ".join(step(k, p) for k, p in steps) for name, steps in config.variables.items()]
    return "\n".join(lines)
