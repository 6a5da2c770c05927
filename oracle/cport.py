"""ctypes front for the plain-C oracle (oracle/c/aggfly_ref.c).  TEST INFRASTRUCTURE.

Gives tests and bench.py's ``cpu_baseline`` leg the numba-engine arithmetic of the
reference (`aggfly/aggregate/nb_kernels.py:121-251`, `aggfly/aggregate/spatial.py:181-186`)
at C speed.  ``build()`` compiles it with gcc; nothing under ``aggfly_amd/`` imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libaggfly_ref.so")
_lib = None

STAT_CODE = {"mean": 0, "sum": 1, "min": 2, "max": 3, "nanmean": 4}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "aggfly_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _sfx(a: np.ndarray) -> str:
    if a.dtype == np.float64:
        return "f64"
    if a.dtype == np.float32:
        return "f32"
    raise TypeError(f"oracle C port handles float32/float64, got {a.dtype}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def block_stat(cube: np.ndarray, bounds: np.ndarray, calc: str) -> np.ndarray:
    cube = np.ascontiguousarray(cube)
    bounds = np.ascontiguousarray(bounds, dtype=np.int64)
    T, NY, NX = cube.shape
    G = len(bounds) - 1
    out = np.empty((G, NY, NX), dtype=cube.dtype)
    getattr(lib(), f"ref_block_stat_{_sfx(cube)}")(
        _p(cube), C.c_int64(T), C.c_int64(NY), C.c_int64(NX), _p(bounds), C.c_int64(G),
        C.c_int(STAT_CODE[calc]), _p(out))
    return out


def _block_thr(name, cube, bounds, ddargs):
    cube = np.ascontiguousarray(cube)
    bounds = np.ascontiguousarray(bounds, dtype=np.int64)
    dda = np.ascontiguousarray(np.atleast_2d(np.asarray(ddargs, dtype=np.float64)))
    T, NY, NX = cube.shape
    G, D = len(bounds) - 1, dda.shape[0]
    out = np.empty((G, NY, NX, D), dtype=cube.dtype)
    getattr(lib(), f"ref_block_{name}_{_sfx(cube)}")(
        _p(cube), C.c_int64(T), C.c_int64(NY), C.c_int64(NX), _p(bounds), C.c_int64(G),
        _p(dda), C.c_int64(D), _p(out))
    return out


def block_dd(cube, bounds, ddargs):
    return _block_thr("dd", cube, bounds, ddargs)


def block_bins(cube, bounds, ddargs):
    return _block_thr("bins", cube, bounds, ddargs)


def block_sine_dd(cube, bounds, ddargs):
    return _block_thr("sine_dd", cube, bounds, ddargs)


def resample(cube, bounds, calc, ddargs=None, multi_dd=False):
    """`numba_resample` (`nb_kernels.py:271-305`) at C speed."""
    if calc in STAT_CODE:
        return block_stat(cube, bounds, calc)
    out = {"dd": block_dd, "bins": block_bins, "sine_dd": block_sine_dd}[calc](cube, bounds, ddargs)
    return out if multi_dd else np.ascontiguousarray(out[..., 0])


def power(x: np.ndarray, e) -> np.ndarray:
    x = np.ascontiguousarray(x)
    out = np.empty_like(x)
    getattr(lib(), f"ref_power_{_sfx(x)}")(_p(x), C.c_int64(x.size), C.c_double(float(e)), _p(out))
    return out


def scatter_block(block, region_idx, cell_idx, w, n_regions):
    block = np.ascontiguousarray(block, dtype=np.float64)
    ri = np.ascontiguousarray(region_idx, dtype=np.int64)
    ci = np.ascontiguousarray(cell_idx, dtype=np.int64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    n_cells, nt = block.shape
    out = np.empty((n_regions, nt), dtype=np.float64)
    lib().ref_scatter_block(_p(block), C.c_int64(n_cells), C.c_int64(nt), _p(ri), _p(ci), _p(w),
                            C.c_int64(len(w)), C.c_int64(n_regions), _p(out))
    return out


def set_threads(n: int):
    os.environ["OMP_NUM_THREADS"] = str(n)
