"""CPU oracle for the aggregate_dataset() hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's algorithm (dylanhogan/aggfly
v0.2.0) for the one path this repository accelerates.  It exists so that the HIP
path can be checked against it; it is never the thing measured or shipped.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``aggfly_amd/`` imports it, and the product
path raises when the HIP extension is missing rather than falling back here.

Parity status: PINNED.  The restatement reproduces every golden vector the
reference's own tests hold for this path (``tests/golden/*.json``, taken from
``aggfly/tests/test_aggregate.py``; see ``tests/test_oracle_golden.py``).  The
reference itself cannot be imported in this container or on the GPU box (it needs
dask/xarray/numba/geopandas and Python >= 3.11, none of which are installed), and
it is pure Python so there is nothing to compile into ``oracle/_ref``.

Arithmetic that lives in third-party dependencies of the reference (pinned in its
``uv.lock``): numpy 2.4.6 (``np.mean/sum/min/max/power/add.at``), numba 0.66.0
(JIT of ``aggfly/aggregate/nb_kernels.py:121-251``), pandas 3.0.3
(``resample(freq).count()``), xarray 2026.7.0 (resample grouping).  Their published
semantics are restated here; where the installed numpy/pandas provide the same
call (``np.add.at``, ``Series.resample().count()``) it is called directly.

Layout
------
``ref_temporal.py``   group bounds + the four grouped reducers (numba kernels and
                      their dask-path twins)          nb_kernels.py, temporal.py
``ref_spatial.py``    COO triplets, scatter-add, divide, NaN-row policy  spatial.py
``ref_aggregate.py``  spec-DSL interpreter, transforms, final merge     aggregate.py
``ref_calendar.py``   noleap / 360_day group bounds without cftime
``c/``                plain-C restatement of the same kernels (gcc, OpenMP over
                      grid rows like numba's prange) for mid-size checks and the
                      CPU baseline timing
"""
