"""aggfly_amd — MI355X-native engine for aggfly's ``aggregate_dataset()`` hot path.

Drop-in for the aggregation half of ``import aggfly as af`` (the re-export list mirrors
`aggfly/__init__.py:1-27` for the names on or next to the hot path).  Execution is by
hand-written HIP kernels for gfx950 behind a C ABI (``include/aggfly_hip.h``); there is no
CPU engine and no fallback.
"""
__version__ = "0.1.0"

from .aggregate import (  # noqa: F401
    SpatialAggregator,
    TemporalAggregator,
    aggregate_dataset,
    aggregate_space,
    aggregate_time,
    distributed_client,
    is_distributed,
    resolve_engine,
    shutdown_dask_client,
    start_dask_client,
)
from .cfcalendar import CFDatetime, CFTimeIndex, cf_range  # noqa: F401
from .dataarray import DataArray  # noqa: F401
from .dataset import Dataset, Grid  # noqa: F401
from .io import dataset_from_path, dataset_to_zarr, zarr_from_path  # noqa: F401
from .weights import (  # noqa: F401
    CropWeights,
    GeoRegions,
    GridWeights,
    PopWeights,
    SecondaryWeights,
    crop_weights_from_path,
    georegions_from_gdf,
    georegions_from_path,
    georegions_from_table,
    pop_weights_from_path,
    secondary_weights_from_path,
    shapefile_info,
    weights_from_feather,
    weights_from_objects,
)
