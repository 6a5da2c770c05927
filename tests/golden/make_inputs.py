"""Regenerates the seeded inputs that belong to the golden vectors in reference_goldens.json.

The reference's tests draw their inputs from seeded numpy generators
(`aggfly/tests/test_aggregate.py:26-53, 443-451, 620-622, 647-648`); numpy's legacy
``RandomState`` and ``default_rng`` streams are stable across versions, so the inputs are
rebuilt here rather than stored.  Importable (used by the tests) and runnable
(``python tests/golden/make_inputs.py`` prints checksums).
"""
import json
import os

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))


def goldens():
    with open(os.path.join(HERE, "reference_goldens.json")) as f:
        return json.load(f)


def dataset_360_inputs():
    """test_aggregate.py:17-53: seed 1216, N(20,15) on 4 x 12h steps, 2x2 grid, lon 0-360."""
    np.random.seed(1216)
    x = np.linspace(0, 360, 3)
    lon = (x[1:] + x[:-1]) / 2
    y = np.linspace(-90, 90, 3)
    lat = (y[1:] + y[:-1]) / 2
    time = pd.date_range("2000-07-01", periods=4, freq="12h")
    arr = np.random.normal(20, 15, (len(time), len(lat), len(lon)))
    return arr, time, lat, lon


def g1_spec():
    """test_aggregate.py:255-271."""
    return dict(
        bins=[("aggregate", {"calc": "mean", "groupby": "date"}),
              ("aggregate", {"calc": "bins", "groupby": "month", "ddargs": [[-99, 20, 0], [20, 99, 0]]})],
        cooling_dday=[("aggregate", {"calc": "dd", "groupby": "date", "ddargs": [20, 99, 0]}),
                      ("aggregate", {"calc": "sum", "groupby": "month"})],
        tavg=[("aggregate", {"calc": "mean", "groupby": "date"}),
              ("transform", {"transform": "power", "exp": np.arange(1, 3)}),
              ("aggregate", {"calc": "sum", "groupby": "month"})],
    )


def g2_spec():
    """test_aggregate.py:300-308."""
    return dict(tavg=g1_spec()["tavg"])


def g2_weights_table():
    g = goldens()["G2_weights"]
    return pd.DataFrame({k: g[k] for k in ("cell_id", "index_right", "weight")})


def cftime_cube(ndays, nan=False, seed=0):
    """test_aggregate.py:443-451: default_rng(seed).normal(15, 12, (ndays, 2, 2)) (+ NaNs)."""
    arr = np.random.default_rng(seed).normal(15, 12, (ndays, 2, 2))
    if nan:
        arr[:, 0, 0] = np.nan
        arr[ndays // 3, 1, 1] = np.nan
    return arr, np.array([-45.0, 45.0]), np.array([10.0, 100.0])


def k3_specs():
    """test_aggregate.py:473-480."""
    return {
        "mean_m": [("aggregate", {"calc": "mean", "groupby": "month"})],
        "sum_y": [("aggregate", {"calc": "sum", "groupby": "year"})],
        "max_m": [("aggregate", {"calc": "max", "groupby": "month"})],
        "dd_m": [("aggregate", {"calc": "dd", "groupby": "month", "ddargs": [10, 30, 0]})],
        "bins_m": [("aggregate", {"calc": "bins", "groupby": "month", "ddargs": [[0, 15, 0], [15, 30, 0]]})],
        "sine_m": [("aggregate", {"calc": "sine_dd", "groupby": "month", "ddargs": [10, 30, 0]})],
    }


def k7_case(which):
    """test_aggregate.py:614-664: spatial cases on a 2x2 grid."""
    lat = np.array([0.0, 1.0]); lon = np.array([0.0, 1.0])
    if which == "multiregion_nan":
        time = pd.date_range("2000-07-01", periods=3, freq="D")
        vals = np.random.default_rng(7).normal(20, 5, (3, 2, 2))
        vals[1, 0, 0] = np.nan
        vals[2, 1, 1] = np.nan
    else:
        time = pd.date_range("2000-07-01", periods=2, freq="D")
        vals = np.random.default_rng(3).normal(20, 5, (2, 2, 2))
        vals[0, 0, 0] = vals[0, 0, 1] = vals[0, 1, 0] = np.nan
    wdf = pd.DataFrame(goldens()["K7_weights"])
    return vals, time, lat, lon, wdf


if __name__ == "__main__":
    arr, *_ = dataset_360_inputs()
    print("dataset_360 sum", repr(float(arr.sum())))
    for c in ("multiregion_nan", "dropna_empty_group"):
        print(c, repr(float(np.nansum(k7_case(c)[0]))))
