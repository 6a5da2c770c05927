"""Weights and regions as the hot path consumes them.

Computing weights (polygon/cell intersection, secondary rasters) stays on the CPU in the
reference's own `aggfly/weights/` + `aggfly/regions/` (geopandas / shapely / rasterio), is
cached there as ``.feather`` (`aggfly/cache/project_cache.py:46-47`), and is OUT OF SCOPE
for this engine (BASELINE.json north_star: "weights stay precomputed/cached on CPU and are
uploaded once as CSR").  What the path needs from those objects is small and is what these
classes hold:

* the weights table — rows ``cell_id``, ``index_right``, ``weight``
  (`aggfly/weights/grid_weights.py:194-196`), ``cell_id`` positional on the clipped,
  +-180-sorted grid (`aggfly/dataset/grid.py:214-217`, `grid_weights.py:642-644`);
* the region-id column of the shapefile, indexed by shapefile row
  (`aggfly/aggregate/aggregate.py:276-280`);
* the zero-weight policy flag (`aggfly/aggregate/spatial.py:69`).
"""
from __future__ import annotations

import copy
from typing import Optional

import numpy as np
import pandas as pd

from .dataset import Dataset, Grid


class GeoRegions:
    """Region table stand-in for `aggfly/regions/georegions.py:22`: ``shp`` is a DataFrame
    (a GeoDataFrame works too) whose index is the region row id used by ``index_right``;
    ``regionid`` names the id column.  Optional ``minx,miny,maxx,maxy`` columns (or a
    GeoDataFrame's ``bounds``) enable clipping a dataset to the regions' extent."""

    def __init__(self, shp: pd.DataFrame, regionid: str = "geoid", name: Optional[str] = None):
        if regionid not in shp.columns:
            raise ValueError(f"regionid column {regionid!r} not in region table")
        self.shp = shp
        self.regionid = regionid
        self.name = name

    @property
    def bounds(self):
        if all(c in self.shp.columns for c in ("minx", "miny", "maxx", "maxy")):
            return self.shp[["minx", "miny", "maxx", "maxy"]].to_numpy(dtype=float)
        b = getattr(self.shp, "bounds", None)
        if b is not None and hasattr(b, "to_numpy"):
            return b.to_numpy(dtype=float)
        raise ValueError("region table carries no bounds (minx, miny, maxx, maxy)")

    @property
    def total_bounds(self):
        b = self.bounds
        return np.array([b[:, 0].min(), b[:, 1].min(), b[:, 2].max(), b[:, 3].max()])


def georegions_from_table(shp: pd.DataFrame, regionid: str = "geoid", name=None) -> GeoRegions:
    return GeoRegions(shp, regionid, name)


class GridWeights:
    """Holder mirroring `aggfly/weights/grid_weights.py:31` for the attributes the
    aggregation path reads: ``grid``, ``georegions``, ``weights``, ``zero_weight``."""

    ZERO_WEIGHT = ("nan", "area", "drop")

    def __init__(self, grid: Grid, georegions: GeoRegions, weights: Optional[pd.DataFrame] = None,
                 raster_weights=None, zero_weight: str = "nan", project_dir=None):
        if zero_weight not in self.ZERO_WEIGHT:
            raise ValueError(f"zero_weight must be one of {self.ZERO_WEIGHT}, got {zero_weight!r}")
        if getattr(grid, "lon_is_360", False):
            raise AssertionError("GridWeights needs a +-180 grid (grid_weights.py:104)")
        self.grid = grid
        self.georegions = georegions
        self.raster_weights = raster_weights
        self.zero_weight = zero_weight
        self.project_dir = project_dir
        self.weights = None
        if weights is not None:
            self.set_table(weights)

    def set_table(self, table: pd.DataFrame):
        missing = [c for c in ("cell_id", "index_right", "weight") if c not in table.columns]
        if missing:
            raise ValueError(f"weights table lacks columns {missing}")
        self.weights = table

    def calculate_weights(self):
        """The reference computes area / secondary weights here (`grid_weights.py:140-213`).
        That geometry pipeline is not part of this engine: supply the table (``table=`` /
        ``set_table``) or a cached ``.feather`` (``weights_from_feather``)."""
        if self.weights is None:
            raise NotImplementedError(
                "weights are precomputed on the CPU by aggfly.weights (geopandas/shapely) and "
                "handed to this engine as a table; pass table=... to weights_from_objects or "
                "load the reference's .feather cache with weights_from_feather().")
        return self.weights


def weights_from_objects(clim: Dataset, georegions: GeoRegions, secondary_weights=None, project_dir=None,
                         table: Optional[pd.DataFrame] = None, **kwargs) -> GridWeights:
    """`weights_from_objects` (`grid_weights.py:614-660`): the grid comes from a copy of the
    dataset rescaled to +-180 (`:642-644`), so ``cell_id`` refers to the longitude-sorted
    grid.  ``table`` carries the precomputed weights."""
    if "default_to_area_weights" in kwargs:
        import warnings
        warnings.warn("default_to_area_weights is deprecated; use zero_weight='area' or 'drop'",
                      DeprecationWarning, stacklevel=2)
        kwargs.setdefault("zero_weight", "area" if kwargs.pop("default_to_area_weights") else "drop")
    zero_weight = kwargs.pop("zero_weight", "nan")
    c = clim.deepcopy()
    if c.lon_is_360:
        c.rescale_longitude()
    return GridWeights(copy.deepcopy(c.grid), georegions, table, secondary_weights, zero_weight, project_dir)


def weights_from_feather(path: str, clim: Dataset, georegions: GeoRegions, zero_weight: str = "nan") -> GridWeights:
    """Read a weights table the reference cached as feather
    (`aggfly/cache/project_cache.py:72-100`, layout ``{project_dir}/tmp/GridWeights/mod-<sha>/<sha>.feather``)."""
    return weights_from_objects(clim, georegions, table=_read_feather(path), zero_weight=zero_weight)


def _read_feather(path: str) -> pd.DataFrame:
    """Feather V2 is the Arrow IPC file format; V1 files go through pyarrow.feather."""
    import pyarrow as pa
    try:
        with pa.OSFile(path, "rb") as f:
            return pa.ipc.open_file(f).read_all().to_pandas()
    except pa.ArrowInvalid:
        import warnings
        import pyarrow.feather as feather
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", FutureWarning)
            return feather.read_feather(path)


def _read_table(path: str) -> pd.DataFrame:
    ext = path.rsplit(".", 1)[-1].lower()
    if ext == "csv":
        return pd.read_csv(path)
    if ext in ("parquet", "pq"):
        return pd.read_parquet(path)
    if ext in ("feather", "arrow"):
        return _read_feather(path)
    raise ValueError(f"unsupported table format: {path}")


def georegions_from_path(path: str, regionid: str = "geoid", region_list=None, name=None) -> GeoRegions:
    """`georegions_from_path` (`aggfly/regions/georegions.py:220-244`).  A region TABLE
    (csv / parquet / feather: the id column, optionally minx, miny, maxx, maxy) is read
    directly; a shapefile needs geopandas, which stays a CPU-side optional dependency."""
    if path.lower().endswith((".shp", ".gpkg", ".geojson", ".json")):
        try:
            import geopandas as gpd
        except ImportError as e:
            raise ImportError(
                f"reading {path} needs geopandas; export the attribute table (id column + bounds) to "
                "csv/parquet, which is all the aggregation path uses of the regions") from e
        shp = gpd.read_file(path)
    else:
        shp = _read_table(path)
    if region_list is not None:
        shp = shp[shp[regionid].astype(str).isin([str(r) for r in region_list])]
    return GeoRegions(shp, regionid, name)


def georegions_from_gdf(gdf, regionid: str = "geoid", region_list=None, name=None) -> GeoRegions:
    """`georegions_from_gdf` (`aggfly/regions/georegions.py:246-270`): any (Geo)DataFrame with the id column."""
    if region_list is not None:
        gdf = gdf[gdf[regionid].astype(str).isin([str(r) for r in region_list])]
    return GeoRegions(gdf, regionid, name)


def _cpu_side_only(name):
    def stub(*args, **kwargs):
        raise NotImplementedError(
            f"{name} belongs to aggfly's CPU-side weights pipeline (geopandas / rasterio), which this engine does not "
            "reimplement: compute the weights with aggfly once, then hand the cached table to weights_from_objects(table=...) "
            "or weights_from_feather().")
    stub.__name__ = name
    return stub


# names of the reference's weights producers (`aggfly/__init__.py:13-22`), present so that
# `import aggfly_amd as af` scripts fail with a pointer instead of an AttributeError
pop_weights_from_path = _cpu_side_only("pop_weights_from_path")
crop_weights_from_path = _cpu_side_only("crop_weights_from_path")
secondary_weights_from_path = _cpu_side_only("secondary_weights_from_path")
shapefile_info = _cpu_side_only("shapefile_info")


def _cpu_side_only_class(name):
    init = _cpu_side_only(name)
    return type(name, (), {"__init__": lambda self, *a, **k: init(*a, **k), "__doc__": f"`{name}` (aggfly/weights/): CPU-side, see the error it raises."})


# ... and of its secondary-weights classes (`aggfly/__init__.py:14-17`): `from aggfly_amd import PopWeights` imports, constructing one points the same way
SecondaryWeights = _cpu_side_only_class("SecondaryWeights")
PopWeights = _cpu_side_only_class("PopWeights")
CropWeights = _cpu_side_only_class("CropWeights")
