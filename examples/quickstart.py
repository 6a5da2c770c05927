#!/usr/bin/env python3
"""End-to-end on synthetic inputs: write a year of hourly "ERA5" to Zarr, a region table and a
weights table, then aggregate through the Python API and through the YAML CLI (needs an MI355X).

    python examples/quickstart.py /tmp/aggfly_demo
"""
import os
import sys

import numpy as np
import pandas as pd
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aggfly_amd as af  # noqa: E402
from aggfly_amd import synth  # noqa: E402
from aggfly_amd.cli.main import cli  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/aggfly_demo"
os.makedirs(out, exist_ok=True)
T, ny, nx = 8760, 104, 236                                  # one year, CONUS window at 0.25 deg
lat, lon = 24.0 + 0.25 * np.arange(ny), 235.0 + 0.25 * np.arange(nx)
cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15)
ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                             {"time": pd.date_range("2001-01-01", periods=T, freq="h"), "latitude": lat, "longitude": lon}),
                lon_is_360=True)
af.dataset_to_zarr(ds, f"{out}/era5_t2m_2001.zarr", var="t2m")
tab = synth.weights_table(ny, nx, 3100, seed=7)
tab.to_parquet(f"{out}/weights.parquet", index=False)
regions = pd.DataFrame({"fips": [f"{i:05d}" for i in range(int(tab.index_right.max()) + 1)]})
regions.to_parquet(f"{out}/regions.parquet", index=False)

# --- Python API (same call shape as `import aggfly as af`) ---
data = af.dataset_from_path(f"{out}/era5_t2m_2001.zarr", var="t2m", preprocess=lambda x: x - 273.15, device="cuda")
w = af.weights_from_objects(data, af.georegions_from_path(f"{out}/regions.parquet", "fips"), table=tab)
panel = af.aggregate_dataset(
    dataset=data, weights=w,
    tavg=[("aggregate", {"calc": "mean", "groupby": "date"}),
          ("transform", {"transform": "power", "exp": np.arange(1, 5)}),
          ("aggregate", {"calc": "sum", "groupby": "year"})],
    gdd=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [10, 30, 0]}),
         ("aggregate", {"calc": "sum", "groupby": "year"})])
print(panel.head())

# --- the YAML CLI ---
cfg = {"regions": {"path": f"{out}/regions.parquet", "regionid": "fips"},
       "dataset": {"path": f"{out}/era5_t2m_{{year}}.zarr", "var": "t2m", "preprocess": "kelvin_to_celsius", "clip_to_regions": False},
       "weights": {"table": f"{out}/weights.parquet"},
       "aggregate": {"engine": "hip", "variables": {"tavg": [["aggregate", {"calc": "mean", "groupby": "date"}],
                                                              ["aggregate", {"calc": "mean", "groupby": "year"}]]}},
       "years": 2001, "output": {"path": f"{out}/panel.csv"}}
with open(f"{out}/config.yaml", "w") as f:
    yaml.safe_dump(cfg, f)
cli(["run", f"{out}/config.yaml"], standalone_mode=False)
print(pd.read_csv(f"{out}/panel.csv").head())
