"""Config loading and validation (`aggfly/cli/config.py:1-400` schema).

Pure: parses YAML into ``RunConfig`` and reports EVERY problem at once (``ConfigError``),
without touching climate data.  Accepted beyond the reference's schema: ``aggregate.engine:
hip`` and ``weights.table`` (path to the precomputed weights table).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

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
class SecondaryWeightsConfig:
    type: str
    path: str
    crop: Optional[str] = None
    feed: Optional[str] = None


@dataclass
class RunConfig:
    regions_path: str
    regionid: str
    region_list: Optional[List[str]]
    dataset_path: str
    var: str
    preprocess: Optional[str]
    preprocess_from: Optional[str]
    lon_is_360: bool
    timecoord: str
    xycoords: Tuple[str, str]
    time_sel: Optional[str]
    chunks: Optional[Dict[str, object]]
    clip_to_regions: bool
    storage_options: Optional[Dict[str, object]]
    reader_engine: Optional[str]
    project_dir: Optional[str]
    weights_table: Optional[str]
    secondary: Optional[SecondaryWeightsConfig]
    zero_weight: str
    engine: str
    variables: Dict[str, List]
    years: Optional[List[int]]
    backend: str
    n_workers: int
    threads_per_worker: int
    output_path: str
    output_format: str

    @property
    def templated(self) -> bool:
        return "{year}" in self.dataset_path

    def resolved_paths(self) -> List[str]:
        if not self.templated:
            return [self.dataset_path]
        return [self.dataset_path.format(year=y) for y in (self.years or [])]

    def to_aggregator_dict(self) -> Dict[str, List]:
        """``variables`` -> the aggregator_dict aggregate_dataset takes.  A transform's ``exp``
        becomes a NumPy array: the library indexes ``exp[0]`` after wrapping non-lists, so a
        bare list [1, 2] would be read as the scalar 1 (`config.py:98-113`)."""
        out = {}
        for name, steps in self.variables.items():
            norm = []
            for step_type, params in steps:
                params = dict(params)
                if step_type == "transform" and "exp" in params:
                    params["exp"] = np.array(params["exp"])
                norm.append((step_type, params))
            out[name] = norm
        return out


def _parse_years(spec, errors):
    if spec is None:
        return None
    if isinstance(spec, bool):
        errors.append("years: must be a range 'start:end', a list, or an int")
        return None
    if isinstance(spec, int):
        return [spec]
    if isinstance(spec, list):
        try:
            return [int(y) for y in spec]
        except (TypeError, ValueError):
            errors.append(f"years: list must contain integers, got {spec!r}")
            return None
    if isinstance(spec, str):
        try:
            if ":" in spec:
                a, b = spec.split(":")
                return list(range(int(a), int(b) + 1))
            return [int(spec)]
        except ValueError:
            errors.append(f"years: could not parse {spec!r} (use 'start:end' or an int)")
            return None
    errors.append(f"years: unsupported type {type(spec).__name__}")
    return None


def _validate_steps(name, steps, errors):
    if not isinstance(steps, list) or not steps:
        errors.append(f"aggregate.variables.{name}: must be a non-empty list of steps")
        return
    fan, conflict = 1, False
    for i, step in enumerate(steps):
        loc = f"aggregate.variables.{name}[{i}]"
        if not (isinstance(step, (list, tuple)) and len(step) == 2):
            errors.append(f"{loc}: each step must be [step_type, params]")
            continue
        step_type, params = step
        if step_type not in ALLOWED_STEP_TYPES:
            errors.append(f"{loc}: unknown step type {step_type!r} (expected one of {sorted(ALLOWED_STEP_TYPES)})")
            continue
        if not isinstance(params, dict):
            errors.append(f"{loc}: params must be a mapping")
            continue
        if step_type == "aggregate":
            calc, groupby = params.get("calc"), params.get("groupby")
            if calc not in ALLOWED_CALCS:
                errors.append(f"{loc}: calc {calc!r} not in {sorted(ALLOWED_CALCS)}")
            if groupby not in ALLOWED_GROUPBY:
                errors.append(f"{loc}: groupby {groupby!r} not in {sorted(ALLOWED_GROUPBY)}")
            if calc in CALCS_NEEDING_DDARGS:
                dd = params.get("ddargs")
                if not isinstance(dd, list) or not dd:
                    errors.append(f"{loc}: calc {calc!r} requires a non-empty 'ddargs' list")
                elif isinstance(dd[0], list) and fan > 1:
                    conflict = True
        else:
            has_exp, has_inter = "exp" in params, "inter" in params
            is_spline = params.get("transform") == "spline" or "spline" in params
            if not (has_exp or has_inter or is_spline):
                errors.append(f"{loc}: transform step needs one of 'exp' (power), 'inter', or transform: spline")
            if has_exp and not isinstance(params["exp"], (list, int)):
                errors.append(f"{loc}: 'exp' must be an int or a list of ints")
            if has_exp and isinstance(params["exp"], list):
                fan = len(params["exp"])
    if conflict:
        errors.append(f"aggregate.variables.{name}: cannot combine a multi-'ddargs' (bins) step with a "
                      "multi-output transform (e.g. multiple exponents) — the library rejects this at runtime")


def parse_config(raw) -> RunConfig:
    errors: List[str] = []
    if raw is None or not isinstance(raw, dict):
        raise ConfigError(["config must be a non-empty YAML mapping"])

    def section(key):
        val = raw.get(key)
        if val is None:
            return {}
        if not isinstance(val, dict):
            errors.append(f"{key}: must be a mapping")
            return {}
        return val

    regions, dataset, weights = section("regions"), section("dataset"), section("weights")
    aggregate, execution, output = section("aggregate"), section("execution"), section("output")

    if not regions.get("path"):
        errors.append("regions.path is required")
    if not regions.get("regionid"):
        errors.append("regions.regionid is required")
    if not dataset.get("path"):
        errors.append("dataset.path is required")
    if not dataset.get("var"):
        errors.append("dataset.var is required")
    preprocess, preprocess_from = dataset.get("preprocess"), dataset.get("preprocess_from")
    if preprocess is not None and preprocess_from is not None:
        errors.append("dataset: set at most one of 'preprocess' and 'preprocess_from'")
    if preprocess_from is not None and ":" not in str(preprocess_from):
        errors.append("dataset.preprocess_from must be 'path/to/file.py:function'")
    xycoords = dataset.get("xycoords", ["longitude", "latitude"])
    if not (isinstance(xycoords, list) and len(xycoords) == 2):
        errors.append("dataset.xycoords must be a 2-item list [lon_name, lat_name]")
        xycoords = ["longitude", "latitude"]
    storage_options = dataset.get("storage_options")
    if storage_options is not None and not isinstance(storage_options, dict):
        errors.append("dataset.storage_options must be a mapping")
        storage_options = None
    reader_engine = dataset.get("engine")
    if reader_engine is not None and not isinstance(reader_engine, str):
        errors.append("dataset.engine must be a string (e.g. 'zarr')")
        reader_engine = None

    zero_weight = weights.get("zero_weight", "nan")
    if zero_weight not in ALLOWED_ZERO_WEIGHT:
        errors.append(f"weights.zero_weight {zero_weight!r} not in {sorted(ALLOWED_ZERO_WEIGHT)}")
        zero_weight = "nan"
    secondary = None
    sraw = weights.get("secondary")
    if sraw is not None:
        if not isinstance(sraw, dict):
            errors.append("weights.secondary must be a mapping")
        else:
            if sraw.get("type") not in ALLOWED_SECONDARY:
                errors.append(f"weights.secondary.type {sraw.get('type')!r} not in {sorted(ALLOWED_SECONDARY)}")
            if not sraw.get("path"):
                errors.append("weights.secondary.path is required")
            secondary = SecondaryWeightsConfig(sraw.get("type"), sraw.get("path"), sraw.get("crop"), sraw.get("feed"))

    engine = aggregate.get("engine", "auto")
    if engine not in ALLOWED_ENGINE:
        errors.append(f"aggregate.engine {engine!r} not in {sorted(ALLOWED_ENGINE)}")
    variables = aggregate.get("variables")
    if not isinstance(variables, dict) or not variables:
        errors.append("aggregate.variables must be a non-empty mapping of name -> steps")
        variables = {}
    else:
        for name, steps in variables.items():
            _validate_steps(name, steps, errors)

    years = _parse_years(raw.get("years"), errors)
    backend = execution.get("backend", "threads")
    if backend not in ALLOWED_BACKEND:
        errors.append(f"execution.backend {backend!r} not in {sorted(ALLOWED_BACKEND)}")

    output_path = output.get("path")
    if not output_path:
        errors.append("output.path is required")
    output_format = output.get("format")
    if output_format is None and output_path:
        ext = os.path.splitext(str(output_path))[1].lstrip(".").lower()
        output_format = {"pq": "parquet"}.get(ext, ext)
    if output_format not in ALLOWED_FORMAT:
        errors.append(f"output.format {output_format!r} not in {sorted(ALLOWED_FORMAT)} "
                      "(set output.format or use a .parquet/.feather/.csv extension)")
    if dataset.get("path") and "{year}" in str(dataset.get("path")) and not years:
        errors.append("dataset.path contains '{year}' but no 'years' were given (add years: 'start:end')")
    if errors:
        raise ConfigError(errors)
    return RunConfig(
        regions_path=regions["path"], regionid=regions["regionid"], region_list=regions.get("region_list"),
        dataset_path=dataset["path"], var=dataset["var"], preprocess=preprocess, preprocess_from=preprocess_from,
        lon_is_360=bool(dataset.get("lon_is_360", True)), timecoord=dataset.get("timecoord", "time"),
        xycoords=(xycoords[0], xycoords[1]), time_sel=dataset.get("time_sel"), chunks=dataset.get("chunks"),
        clip_to_regions=bool(dataset.get("clip_to_regions", True)), storage_options=storage_options,
        reader_engine=reader_engine, project_dir=weights.get("project_dir"), weights_table=weights.get("table"),
        secondary=secondary, zero_weight=zero_weight, engine=engine, variables=variables, years=years,
        backend=backend, n_workers=int(execution.get("n_workers", 1)),
        threads_per_worker=int(execution.get("threads_per_worker", 1)),
        output_path=output_path, output_format=output_format)


def load_config(path) -> RunConfig:
    try:
        with open(path) as f:
            raw = yaml.safe_load(f)
    except FileNotFoundError:
        raise ConfigError([f"config file not found: {path}"])
    except yaml.YAMLError as e:
        raise ConfigError([f"could not parse YAML: {e}"])
    return parse_config(raw)
