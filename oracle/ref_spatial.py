"""Oracle: sparse region x cell weighted average.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates `aggfly/aggregate/spatial.py:71-199`:

* `_weight_triplets` (`:157-178`): COO rows = position of ``index_right`` in the sorted
  unique region ids, COO cols = position of ``cell_id`` in the stacked cell order, entries
  whose cell is absent from the climate grid are dropped;
* shared validity (`:114-119`): a (cell, time) is used only if every output name is
  non-NaN there;
* `_scatter_block` (`:181-186`): ``contrib = w[:,None]*block[cell_idx,:]`` then
  ``np.add.at`` into (n_regions, t) — float64, summed in COO (table) order;
* divide (`:127-133`): ``num/den`` where ``den != 0`` else NaN;
* long frame + NaN-row policy (`:136-154`).
"""
from __future__ import annotations

import numpy as np
import pandas as pd


def weight_triplets(wdf: pd.DataFrame, cell_ids: np.ndarray):
    """`_weight_triplets` `spatial.py:157-178`."""
    cellpos = {int(c): i for i, c in enumerate(cell_ids)}
    region_ids = np.sort(wdf["index_right"].unique())
    regionpos = {r: i for i, r in enumerate(region_ids)}
    rows = wdf["index_right"].map(regionpos).to_numpy()
    cols = wdf["cell_id"].map(cellpos).to_numpy()
    keep = ~pd.isna(cols)
    return (rows[keep].astype(np.intp), cols[keep].astype(np.intp),
            wdf["weight"].to_numpy(dtype=float)[keep], region_ids)


def scatter_block(block, region_idx, cell_idx, w_vals, n_regions):
    """`_scatter_block` `spatial.py:181-186` on block[n_cells, t]."""
    contrib = w_vals[:, None] * block[cell_idx, :]
    out = np.zeros((n_regions, block.shape[1]), dtype=float)
    np.add.at(out, region_idx, contrib)
    return out


def spatial_num_den(arrs: dict, wdf: pd.DataFrame, cell_ids: np.ndarray):
    """num[name] (R, T'), den (R, T'), region_ids — `spatial.py:103-125`.

    ``arrs[name]`` is (n_cells, T') float64 in ``cell_ids`` order.
    """
    names = list(arrs)
    region_idx, cell_idx, w_vals, region_ids = weight_triplets(wdf, cell_ids)
    n_regions = len(region_ids)
    valid = None
    for nm in names:
        v = ~np.isnan(arrs[nm])
        valid = v if valid is None else (valid & v)
    kw = dict(region_idx=region_idx, cell_idx=cell_idx, w_vals=w_vals, n_regions=n_regions)
    den = scatter_block(valid.astype(float), **kw)
    nums = {nm: scatter_block(np.where(valid, arrs[nm], 0.0), **kw) for nm in names}
    return nums, den, region_ids


def spatial_compute(arrs: dict, time, wdf: pd.DataFrame, cell_ids: np.ndarray,
                    zero_weight: str = "area") -> pd.DataFrame:
    """`SpatialAggregator.compute` `spatial.py:71-154` -> long frame
    [region_id, time, <names>]."""
    names = list(arrs)
    nums, den, region_ids = spatial_num_den(arrs, wdf, cell_ids)
    n_regions, n_time = den.shape
    with np.errstate(invalid="ignore", divide="ignore"):
        res = {nm: np.divide(nums[nm], den, out=np.full_like(den, np.nan), where=den != 0)
               for nm in names}
    time = np.asarray(time, dtype=object) if not isinstance(time, (pd.DatetimeIndex, np.ndarray)) else np.asarray(time)
    out = pd.DataFrame({
        "region_id": np.repeat(region_ids, n_time),
        "time": np.tile(time, n_regions),
    })
    for nm in names:
        out[nm] = res[nm].reshape(-1)
    if zero_weight == "nan":
        wsum = wdf.groupby("index_right")["weight"].sum()
        zero_regions = set(wsum.index[~(wsum > 0)])
        keep = out["region_id"].isin(zero_regions) | out[names].notna().all(axis=1)
        out = out.loc[keep].reset_index(drop=True)
    else:
        out = out.dropna(subset=names).reset_index(drop=True)
    return out


def wavg_loops(vals: dict, time, grid_cell_ids, wdf: pd.DataFrame, names):
    """Independent pure-loop weighted average, the shape of the reference tests' own
    checker (`aggfly/tests/test_aggregate.py:578-601`): used to cross-check
    ``spatial_compute`` itself."""
    cellpos = {int(c): i for i, c in enumerate(grid_cell_ids)}
    rows = []
    for r in np.sort(wdf["index_right"].unique()):
        sub = wdf[wdf["index_right"] == r]
        cidx = sub["cell_id"].map(cellpos).to_numpy()
        wv = sub["weight"].to_numpy(dtype=float)
        for ti in range(len(time)):
            ok = np.ones(len(cidx), bool)
            for nm in names:
                ok &= ~np.isnan(vals[nm][cidx, ti])
            den = wv[ok].sum()
            if den == 0:
                continue
            row = {"region_id": r, "time": time[ti]}
            for nm in names:
                row[nm] = (wv[ok] * vals[nm][cidx, ti][ok]).sum() / den
            rows.append(row)
    return pd.DataFrame(rows)
