"""Synthetic inputs of the shapes BASELINE.json names.

No ERA5/CMIP6 files or shapefiles exist offline, and the weights producer
(`aggfly/weights/grid_weights.py`) needs geopandas, so benches and tests synthesise

* temperatures: ``15 + 12 sin(2 pi doy/365) + 6 sin(2 pi hour/24) + N(0, 3)`` in deg C,
  optionally with a time-invariant NaN "ocean" mask and scattered NaNs;
* a region x cell weights table with the reference's schema (``cell_id``, ``index_right``,
  ``weight``; `grid_weights.py:194-196`): a seeded recursive k-d split of the ny x nx grid
  into R rectangles, interior cells weighted cos(lat), cells on a shared edge given to
  both neighbours with fractions u and 1-u, optional log-normal secondary weights
  normalised per region like `grid_weights.py:487-489`, and a few zero-weight regions.
"""
from __future__ import annotations

import numpy as np
import pandas as pd


def temperature_cube(T: int, ny: int, nx: int, dtype=np.float64, seed: int = 20260101,
                     steps_per_day: int = 24, ocean_frac: float = 0.0, scattered_nan: int = 0,
                     chunk: int = 512) -> np.ndarray:
    """(T, ny, nx) time-major cube."""
    rng = np.random.default_rng(seed)
    out = np.empty((T, ny, nx), dtype=dtype)
    lat_amp = np.linspace(0.6, 1.4, ny)[:, None]
    for k0 in range(0, T, chunk):
        k1 = min(T, k0 + chunk)
        k = np.arange(k0, k1)
        doy = (k // steps_per_day) % 365
        hour = (k % steps_per_day) * (24.0 / steps_per_day)
        base = 15.0 + 12.0 * np.sin(2 * np.pi * doy / 365.0) + 6.0 * np.sin(2 * np.pi * hour / 24.0)
        blk = base[:, None, None] * lat_amp[None] + rng.normal(0.0, 3.0, (k1 - k0, ny, nx))
        out[k0:k1] = blk.astype(dtype)
    if ocean_frac > 0:
        mask = np.random.default_rng(seed + 1).random((ny, nx)) < ocean_frac
        out[:, mask] = np.nan
    if scattered_nan:
        r = np.random.default_rng(seed + 2)
        out[r.integers(0, T, scattered_nan), r.integers(0, ny, scattered_nan), r.integers(0, nx, scattered_nan)] = np.nan
    return out


def _kd_rects(ny, nx, R, rng):
    """Recursive k-d split: always the LARGEST rectangle (the earliest one among equals) is cut along its longer side at a
    random interior position.  A heap keyed on (-area, insertion order) picks exactly the rectangle a linear scan for the
    first maximum would, in O(R log R) instead of O(R^2) (40,000 regions on the 0.1 degree grid took a minute)."""
    import heapq
    heap, alive, n = [(-(ny * nx), 0)], {0: (0, ny, 0, nx)}, 1
    while len(alive) < R:
        _, i = heapq.heappop(heap)
        y0, y1, x0, x1 = alive.pop(i)
        h, w = y1 - y0, x1 - x0
        parts = None
        if h * w >= 2:
            if h >= w and h >= 2:
                c = int(rng.integers(y0 + 1, y1))
                parts = [(y0, c, x0, x1), (c, y1, x0, x1)]
            elif w >= 2:
                c = int(rng.integers(x0 + 1, x1))
                parts = [(y0, y1, x0, c), (y0, y1, c, x1)]
        if parts is None:                      # nothing left to split: the rectangle goes back (to the end, as before)
            alive[n] = (y0, y1, x0, x1)
            break
        for r in parts:
            alive[n] = r
            heapq.heappush(heap, (-((r[1] - r[0]) * (r[3] - r[2])), n))
            n += 1
    return [alive[k] for k in sorted(alive)]


def skewed_weights_table(ny: int, nx: int, R: int, seed: int = 7, sigma: float = 2.0, longest: int = 100_000,
                         lat0: float = 24.0, dlat: float = 0.25) -> pd.DataFrame:
    """A weights table whose rows span four decades of lengths, like real admin-2 / GADM regions on a fine grid (a k-d
    split gives near-uniform rows): region sizes are log-normal (``sigma`` in log space), the largest region is at least
    ``longest`` cells (or a third of the grid), every region holds at least one cell, and the sizes add up to the grid.
    Regions are runs of cells in row-major order (bands and pieces of bands); the last cell of a run also belongs to the
    next region with a random fraction, like a border cell.  Same schema and ordering as `weights_table`."""
    rng = np.random.default_rng(seed)
    C = ny * nx
    R = min(R, C)
    raw = rng.lognormal(0.0, sigma, R)
    big = int(np.argmax(raw))
    want_big = min(longest, C // 3)
    rest = C - want_big - (R - 1)                       # cells left once every other region has its first cell
    others = np.delete(raw, big)
    share = np.floor(others / others.sum() * max(rest, 0)).astype(np.int64)
    sizes = np.ones(R, dtype=np.int64)
    sizes[np.arange(R) != big] += share
    sizes[big] = C - (sizes.sum() - 1)                   # the remainder: >= want_big
    order = rng.permutation(R)                           # where each region sits along the grid
    starts = np.concatenate([[0], np.cumsum(sizes[order])])
    coslat = np.cos(np.deg2rad(lat0 + dlat * np.arange(ny)))
    cell = np.arange(C, dtype=np.int64)
    reg = np.repeat(order, sizes[order]).astype(np.int64)
    wt = coslat[cell // nx].copy()
    # border cells: the first cell of the next run also gets a share of this region
    nxt = starts[1:-1]
    u = rng.uniform(0.05, 0.95, len(nxt))
    df = pd.DataFrame({"cell_id": np.concatenate([cell, nxt]), "index_right": np.concatenate([reg, order[:-1]]).astype(np.int64),
                       "weight": np.concatenate([wt, coslat[nxt // nx] * u])})
    return df.sort_values(["index_right", "cell_id"], kind="stable").reset_index(drop=True)


def weights_table(ny: int, nx: int, R: int, seed: int = 7, secondary: bool = False,
                  zero_frac: float = 0.0, lat0: float = 24.0, dlat: float = 0.25, skew: str = None) -> pd.DataFrame:
    """Weights table [cell_id, index_right, weight] over an ny x nx grid with ~R regions.  ``skew="lognormal"``: row
    lengths over four decades with one row of >= 10^5 cells (`skewed_weights_table`) instead of the near-uniform k-d split."""
    if skew == "lognormal":
        df = skewed_weights_table(ny, nx, R, seed=seed, lat0=lat0, dlat=dlat)
        return _secondary_and_zero(df, ny, nx, seed, secondary, zero_frac, int(df["index_right"].max()) + 1)
    if skew is not None:
        raise ValueError("skew must be None or 'lognormal'")
    rng = np.random.default_rng(seed)
    rects = _kd_rects(ny, nx, R, rng)
    coslat = np.cos(np.deg2rad(lat0 + dlat * np.arange(ny)))
    cell, reg, wt = [], [], []
    for r, (y0, y1, x0, x1) in enumerate(rects):
        ys, xs = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
        w = np.repeat(coslat[y0:y1, None], x1 - x0, axis=1).copy()
        # shared edges: the last row/column also belongs to the next rectangle with fraction 1-u
        cell.append((ys * nx + xs).ravel()); reg.append(np.full(ys.size, r)); wt.append(w.ravel())
        if y1 < ny:  # give the row below a fractional share
            u = rng.uniform(0.05, 0.95, x1 - x0)
            cell.append((y1 * nx + np.arange(x0, x1))); reg.append(np.full(x1 - x0, r)); wt.append(coslat[y1] * u)
    df = pd.DataFrame({"cell_id": np.concatenate(cell).astype(np.int64),
                       "index_right": np.concatenate(reg).astype(np.int64),
                       "weight": np.concatenate(wt)})
    return _secondary_and_zero(df, ny, nx, seed, secondary, zero_frac, len(rects))


def _secondary_and_zero(df, ny, nx, seed, secondary, zero_frac, n_regions):
    if secondary:
        pop = np.random.default_rng(seed + 11).lognormal(0.0, 1.5, ny * nx)
        raster = pop[df["cell_id"].to_numpy()]
        area = df["weight"].to_numpy()
        tot = pd.Series(raster).groupby(df["index_right"].to_numpy()).transform("sum").to_numpy()
        df["weight"] = area * raster / tot                      # grid_weights.py:487-489
    if zero_frac > 0:
        nz = max(1, int(zero_frac * n_regions))
        zr = np.random.default_rng(seed + 12).choice(n_regions, nz, replace=False)
        df.loc[df["index_right"].isin(zr), "weight"] = 0.0
    # table order as a geodataframe join would give it: by region, then cell
    return df.sort_values(["index_right", "cell_id"], kind="stable").reset_index(drop=True)


def hourly_bounds(T: int, steps_per_day: int = 24):
    """Inner (daily) bounds over time steps for a regular series without gaps."""
    n_days = (T + steps_per_day - 1) // steps_per_day
    b = np.minimum(np.arange(n_days + 1, dtype=np.int64) * steps_per_day, T)
    return b
