"""ctypes binding of ``libaggfly_hip.so`` (C ABI: include/aggfly_hip.h).

This is the only door from Python into the engine.  There is no CPU fallback: if the
shared library is missing or no GPU is visible, calls raise ``HipEngineError`` — the
reference's numba/dask engines (`aggfly/aggregate/nb_kernels.py`, `temporal.py`) are not
reimplemented on the host.

Device memory, streams and (for multi-GPU) the process group come from PyTorch-ROCm;
only raw pointers and sizes cross into the library.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGGFLY_HIP_LIB") or os.path.join(_HERE, "libaggfly_hip.so")     # override: A/B of two builds

# codes (include/aggfly_hip.h)
F32, F64 = 0, 1
MEAN, SUM, MIN, MAX, NANMEAN, DD, BINS, SINE_DD, IDENTITY = range(9)
TF_NONE, TF_POW, TF_HINGE, TF_INTER = 0, 1, 2, 3
ROUND_INNER, ROUND_HINGE, ROUND_FINAL = 1, 2, 4
CALC_CODE = {"mean": MEAN, "sum": SUM, "min": MIN, "max": MAX, "nanmean": NANMEAN,
             "dd": DD, "bins": BINS, "sine_dd": SINE_DD}
E_INVALID, E_HIP, E_UNSUPPORTED, E_NOMEM = -1, -2, -3, -4
ABI_VERSION = 4                 # AFHIP_ABI_VERSION of include/aggfly_hip.h this binding was written against

EXPORTS = (
    "afhip_last_error", "afhip_abi_version", "afhip_build_info", "afhip_device_count", "afhip_device_info",
    "afhip_group_stat", "afhip_group_dd", "afhip_group_bins", "afhip_group_sine_dd",
    "afhip_csr_create", "afhip_csr_destroy", "afhip_scatter_block", "afhip_spatial_wavg", "afhip_place_box",
    "afhip_plan_create", "afhip_plan_destroy", "afhip_plan_workspace_bytes", "afhip_plan_run_workspace_bytes",
    "afhip_plan_describe", "afhip_plan_run_temporal", "afhip_plan_run",
    "afhip_plan_profile_begin", "afhip_plan_profile_end", "afhip_plan_bind_inter", "afhip_csr_device", "afhip_plan_device", "afhip_transform", "afhip_panel_divide", "afhip_lz4_decode_streams", "afhip_unshuffle_blocks", "afhip_read_probe",
)


class HipEngineError(RuntimeError):
    """The HIP engine is unavailable or a HIP call failed."""


class HipUnsupported(HipEngineError):
    """A valid plan the fused kernels cannot express in one pass (AFHIP_E_UNSUPPORTED)."""


class Column(C.Structure):
    _fields_ = [("inner", C.c_int32), ("transform", C.c_int32), ("outer", C.c_int32),
                ("rounding", C.c_int32), ("inner_args", C.c_double * 3),
                ("transform_arg", C.c_double), ("outer_args", C.c_double * 3)]


class PlanDesc(C.Structure):
    _fields_ = [("T", C.c_int64), ("n_cells", C.c_int64), ("dtype", C.c_int32), ("K", C.c_int32),
                ("G1", C.c_int64), ("inner_bounds", C.POINTER(C.c_int64)),
                ("P", C.c_int64), ("outer_bounds", C.POINTER(C.c_int64)),
                ("columns", C.POINTER(Column)), ("exact_order", C.c_int32), ("tuning", C.c_int32)]


_lib = None


def load():
    """Load the library (once).  Raises HipEngineError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipEngineError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C aggfly_amd/csrc`).  There is no CPU fallback.")
    try:
        # torch ships its own HIP runtime (same SONAME as /opt/rocm's).  Whichever is loaded first serves the
        # whole process; with the system one first, torch later reports "No HIP GPUs are available".
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - C / ctypes-only consumers
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise HipEngineError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    lib.afhip_last_error.restype = C.c_char_p
    lib.afhip_plan_workspace_bytes.restype = i64
    lib.afhip_plan_run_workspace_bytes.restype = i64
    lib.afhip_plan_run_workspace_bytes.argtypes = [vp, vp]
    if lib.afhip_abi_version() != ABI_VERSION:
        raise HipEngineError(f"{LIB_PATH} speaks ABI {lib.afhip_abi_version()}, this binding ABI {ABI_VERSION}: rebuild it (make -C aggfly_amd/csrc)")
    lib.afhip_csr_destroy.restype = None
    lib.afhip_plan_destroy.restype = None
    lib.afhip_build_info.argtypes = [C.c_char_p, i32]
    lib.afhip_read_probe.argtypes = [vp, i64, i64, i32, C.POINTER(C.c_float), vp]
    lib.afhip_device_info.argtypes = [i32, C.c_char_p, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64)]
    lib.afhip_group_stat.argtypes = [vp, i32, i64, i64, vp, i64, i32, vp, vp]
    for nm in ("afhip_group_dd", "afhip_group_bins", "afhip_group_sine_dd"):
        getattr(lib, nm).argtypes = [vp, i32, i64, i64, vp, i64, vp, i64, vp, vp]
    lib.afhip_csr_create.argtypes = [vp, vp, vp, i64, i64, i64, C.POINTER(vp)]
    lib.afhip_csr_destroy.argtypes = [vp]
    lib.afhip_scatter_block.argtypes = [vp, vp, i64, vp, vp]
    lib.afhip_place_box.argtypes = [vp, vp, i32] + [i64] * 13 + [vp]
    lib.afhip_spatial_wavg.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp]
    lib.afhip_plan_create.argtypes = [C.POINTER(PlanDesc), C.POINTER(vp)]
    lib.afhip_plan_destroy.argtypes = [vp]
    lib.afhip_plan_workspace_bytes.argtypes = [vp]
    lib.afhip_plan_describe.argtypes = [vp, C.c_char_p, i32]
    lib.afhip_plan_run_temporal.argtypes = [vp, vp, vp, vp, i64, vp]
    lib.afhip_plan_run.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, C.POINTER(C.c_float)]
    lib.afhip_plan_bind_inter.argtypes = [vp, i32, vp, i32]
    lib.afhip_transform.argtypes = [vp, i32, i64, i32, dbl, vp, i32, vp, i32, vp]
    lib.afhip_panel_divide.argtypes = [vp, vp, vp, i64, i64, i64, vp]
    lib.afhip_lz4_decode_streams.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    lib.afhip_unshuffle_blocks.argtypes = [vp, vp, vp, i64, i32, vp]
    lib.afhip_plan_profile_begin.argtypes = [vp, i64]
    lib.afhip_plan_profile_end.argtypes = [vp, C.POINTER(C.c_float), i64]
    lib.afhip_plan_profile_end.restype = i64
    lib.afhip_csr_device.argtypes = [vp]
    lib.afhip_plan_device.argtypes = [vp]
    _lib = lib
    return lib


def _check(rc: int):
    if rc == 0:
        return
    msg = load().afhip_last_error().decode("utf-8", "replace")
    if rc == E_INVALID:
        raise ValueError(msg)
    if rc == E_UNSUPPORTED:
        raise HipUnsupported(msg)
    raise HipEngineError(f"HIP engine error {rc}: {msg}")


def device_count() -> int:
    return int(load().afhip_device_count())


def build_info() -> dict:
    """What the loaded build holds: {"menu": "full" | "arms" | "dev", "variants", "arms", "region_fused_twins", "abi"} (`afhip_build_info`)."""
    buf = C.create_string_buffer(256)
    load().afhip_build_info(buf, 256)
    out = dict(kv.split("=") for kv in buf.value.decode().split())
    return {k: (int(v) if v.isdigit() else v) for k, v in out.items()}


def device_info(dev: int = 0) -> dict:
    name, arch = C.create_string_buffer(256), C.create_string_buffer(256)
    cus, mem = C.c_int(0), C.c_int64(0)
    _check(load().afhip_device_info(dev, name, 256, arch, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "arch": arch.value.decode(), "cus": cus.value, "hbm_bytes": mem.value}


def read_probe(cube, launches: int = 10):
    """`afhip_read_probe`: per-launch ms of a bare streaming read of ``cube`` ([T, ...] HBM tensor, rows of a multiple of 8 bytes)
    with the temporal kernels' access pattern — this box's read ceiling for the shape."""
    require_gpu()
    cube, T, n_cells = _dev_cube(cube)
    ms = (C.c_float * int(launches))()
    _check(load().afhip_read_probe(cube.data_ptr(), T, n_cells * cube.element_size(), int(launches), ms, _stream_ptr(cube)))
    return [float(v) for v in ms]


def place_box(chunk, cube, box_in_chunk, at):
    """cube[t0:t0+nt, y0:y0+ny, x0:x0+nx] = chunk[st:st+nt, sy:sy+ny, sx:sx+nx] on the current stream: the
    ingestion route's device-side scatter.  Both tensors contiguous, same dtype (2 / 4 / 8-byte elements)."""
    (st, sy, sx, nt, ny, nx), (t0, y0, x0) = box_in_chunk, at
    _check(load().afhip_place_box(chunk.data_ptr(), cube.data_ptr(), chunk.element_size(), chunk.shape[1], chunk.shape[2],
                                  st, sy, sx, nt, ny, nx, cube.shape[1], cube.shape[2], t0, y0, x0, _stream_ptr(cube)))


def lz4_decode_streams(comp, streams, n_streams: int, max_dsize: int, tmp, out, errors):
    """`afhip_lz4_decode_streams`: decode ``n_streams`` LZ4 streams (records in the uint8 HBM tensor ``streams``, planned
    by `codec.blosc_lz4_plan`) of the compressed bytes ``comp`` into ``tmp`` / ``out`` on the current stream; malformed
    streams bump the int32 HBM counter ``errors``."""
    _check(load().afhip_lz4_decode_streams(comp.data_ptr(), streams.data_ptr(), int(n_streams), int(max_dsize),
                                           tmp.data_ptr() if tmp is not None else None, out.data_ptr(), errors.data_ptr(), _stream_ptr(comp)))


def unshuffle_blocks(tmp, out, blocks, n_blocks: int, max_bsize: int):
    """`afhip_unshuffle_blocks`: Blosc's byte shuffle undone per block (records in the uint8 HBM tensor ``blocks``)."""
    _check(load().afhip_unshuffle_blocks(tmp.data_ptr(), out.data_ptr(), blocks.data_ptr(), int(n_blocks), int(max_bsize), _stream_ptr(out)))


def require_gpu():
    """Fail loudly when the product path is asked to run without a GPU."""
    if device_count() < 1:
        raise HipEngineError("no HIP device visible: the aggregation engine runs on MI355X only (no CPU fallback)")


# --------------------------------------------------------------------------------------
# torch plumbing (device memory + stream only)
# --------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _stream_ptr(t=None):
    """The current stream of the device that holds tensor ``t`` (of the current device when ``t`` is None).  A worker thread
    starts on device 0 whatever its parent had selected; work on a tensor must go to a stream of the tensor's own device."""
    torch = _torch()
    dev = t.device if t is not None and getattr(t, "is_cuda", False) else None
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _device_index(device=None) -> int:
    """Index of a torch device / tensor's device / int; None -> the calling thread's current device."""
    torch = _torch()
    if device is None:
        return int(torch.cuda.current_device())
    if hasattr(device, "device"):
        device = device.device
    if isinstance(device, int):
        return device
    device = torch.device(device)
    return int(torch.cuda.current_device()) if device.index is None else int(device.index)


class _on_device:
    """`with _on_device(i):` makes device ``i`` current for the calling thread (handles belong to the device that is current
    when the library creates them, include/aggfly_hip.h "Devices") and restores the previous one."""

    def __init__(self, index: int):
        self.index, self.ctx = int(index), None

    def __enter__(self):
        torch = _torch()
        if int(torch.cuda.current_device()) != self.index:
            self.ctx = torch.cuda.device(self.index)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _dtype_code(t) -> int:
    torch = _torch()
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise TypeError(f"climate cube must be float32 or float64, got {t.dtype}")


def _dev_cube(t):
    """(T, cells...) CUDA tensor, contiguous, time-major -> (tensor, T, n_cells)."""
    if not t.is_cuda:
        raise HipEngineError("cube must be resident in HBM (a CUDA/HIP tensor)")
    t = t.contiguous()
    T = t.shape[0]
    return t, int(T), int(t.numel() // max(T, 1)) if T else int(np.prod(t.shape[1:]))


def _i64(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


def _group(fn_name, cube, bounds, code=None, ddargs=None):
    torch = _torch()
    lib = load()
    require_gpu()
    cube, T, n_cells = _dev_cube(cube)
    bounds = _i64(bounds)
    G = len(bounds) - 1
    if bounds[0] != 0 or bounds[-1] != T or np.any(np.diff(bounds) < 0):
        raise ValueError("bounds must be monotone and run from 0 to T")
    spatial = tuple(cube.shape[1:])
    if ddargs is None:
        out = torch.empty((G,) + spatial, dtype=cube.dtype, device=cube.device)
        if G:
            _check(lib.afhip_group_stat(cube.data_ptr(), _dtype_code(cube), T, n_cells, bounds.ctypes.data, G,
                                        int(code), out.data_ptr(), _stream_ptr(cube)))
        return out
    dda = np.ascontiguousarray(np.atleast_2d(np.asarray(ddargs, dtype=np.float64)))
    D = dda.shape[0]
    out = torch.empty((G,) + spatial + (D,), dtype=cube.dtype, device=cube.device)
    if G:      # any D: the library runs passes of 16 thresholds inside the call (like the reference's loop over ddargs rows)
        _check(getattr(lib, fn_name)(cube.data_ptr(), _dtype_code(cube), T, n_cells, bounds.ctypes.data, G,
                                     dda.ctypes.data, D, out.data_ptr(), _stream_ptr(cube)))
    return out


def transform(x, kind: str, arg: float = 0.0, other=None, out_dtype=None):
    """Element-wise transform of an HBM tensor by the library's `k_transform` (`afhip_transform`): ``kind`` 'pow' (arg =
    exponent; `np.power`, `aggfly/dataset/dataset.py:527-543`), 'hinge' (arg = knot; `dataset.py:475-481`) or 'inter'
    (``other``: same-shape tensor; `np.multiply`, `dataset.py:547-563`).  -> a new tensor of ``out_dtype`` (default: x's)."""
    torch = _torch()
    lib = load()
    require_gpu()
    if not x.is_cuda:
        raise HipEngineError("transform: the array must be resident in HBM (a CUDA/HIP tensor)")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    code = {"pow": TF_POW, "hinge": TF_HINGE, "inter": TF_INTER}[kind]
    optr, ocode = None, 0
    if code == TF_INTER:
        if other is None or tuple(other.shape) != tuple(x.shape) or not other.is_cuda:
            raise ValueError("transform('inter'): needs a second HBM tensor of the same shape")
        other = other.contiguous()
        optr, ocode = other.data_ptr(), _dtype_code(other)
    _check(lib.afhip_transform(x.data_ptr(), _dtype_code(x), x.numel(), code, float(arg), optr, ocode,
                               out.data_ptr(), _dtype_code(out), _stream_ptr(x)))
    return out


def panel_divide(num, den, out=None):
    """res = num / den where den != 0 else NaN (`aggfly/aggregate/spatial.py:127-133`): num [K, R, P], den [R, P] float64 HBM."""
    torch = _torch()
    require_gpu()
    num, den = num.contiguous(), den.contiguous()
    if num.dtype != torch.float64 or den.dtype != torch.float64 or num.ndim != 3 or tuple(num.shape[1:]) != tuple(den.shape) or not num.is_cuda:
        raise ValueError("panel_divide: num [K, R, P] and den [R, P] must be float64 HBM tensors")
    res = torch.empty_like(num) if out is None else out
    K, R, P = (int(v) for v in num.shape)
    _check(load().afhip_panel_divide(num.data_ptr(), den.data_ptr(), res.data_ptr(), K, R, P, _stream_ptr(num)))
    return res


def group_stat(cube, bounds, calc: str):
    """`_block_stat` (`aggfly/aggregate/nb_kernels.py:121-155`): cube[T,NY,NX] -> out[G,NY,NX]."""
    return _group("afhip_group_stat", cube, bounds, code=CALC_CODE[calc])


def group_dd(cube, bounds, ddargs):
    """`_block_dd` (`nb_kernels.py:158-179`): -> out[G,NY,NX,D]."""
    return _group("afhip_group_dd", cube, bounds, ddargs=ddargs)


def group_bins(cube, bounds, ddargs):
    """`_block_bins` (`nb_kernels.py:182-199`): -> out[G,NY,NX,D]."""
    return _group("afhip_group_bins", cube, bounds, ddargs=ddargs)


def group_sine_dd(cube, bounds, ddargs):
    """`_block_sine_dd` (`nb_kernels.py:202-251`): -> out[G,NY,NX,D]."""
    return _group("afhip_group_sine_dd", cube, bounds, ddargs=ddargs)


class CSR:
    """Region x cell weights resident in HBM (`_weight_triplets`, `aggfly/aggregate/spatial.py:157-178`).

    Built from COO triplets in table order; a stable sort by region row keeps the table's
    entry order inside each row, which is the order `np.add.at` sums in.
    """

    def __init__(self, region_idx, cell_idx, w_vals, n_regions: int, n_cells: int, device=None):
        """``device``: the GPU the tables are uploaded to (torch device / index / a tensor on it; default: the calling
        thread's current device).  Only arrays of that device may be multiplied with this handle."""
        lib = load()
        require_gpu()
        self.device_index = _device_index(device)
        region_idx = _i64(region_idx)
        cell_idx = _i64(cell_idx)
        w = np.ascontiguousarray(np.asarray(w_vals, dtype=np.float64))
        if not (len(region_idx) == len(cell_idx) == len(w)):
            raise ValueError("COO triplets must have equal lengths")
        if len(region_idx) and (region_idx.min() < 0 or region_idx.max() >= n_regions):
            raise ValueError("region index out of range")
        order = np.argsort(region_idx, kind="stable")
        counts = np.bincount(region_idx, minlength=n_regions)
        self.indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.cols = np.ascontiguousarray(cell_idx[order])
        self.w = np.ascontiguousarray(w[order])
        self.R, self.nnz, self.n_cells = int(n_regions), int(len(w)), int(n_cells)
        h = C.c_void_p()
        with _on_device(self.device_index):
            _check(lib.afhip_csr_create(self.indptr.ctypes.data, self.cols.ctypes.data, self.w.ctypes.data,
                                        self.R, self.nnz, self.n_cells, C.byref(h)))
        self._h = h

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            load().afhip_csr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scatter_block(self, block):
        """`_scatter_block` (`spatial.py:181-186`): block[n_cells, t] f64 -> out[R, t]."""
        torch = _torch()
        block = block.contiguous()
        if block.dtype != torch.float64 or block.shape[0] != self.n_cells:
            raise ValueError("block must be float64 [n_cells, t]")
        nt = int(block.shape[1])
        out = torch.empty((self.R, nt), dtype=torch.float64, device=block.device)
        _check(load().afhip_scatter_block(self._h, block.data_ptr(), nt, out.data_ptr(), _stream_ptr(block)))
        return out

    def wavg(self, x):
        """Shared validity + num/den + divide (`spatial.py:110-133`): x[K, n_cells, t] f64
        -> (num[K,R,t], den[R,t], res[K,R,t])."""
        torch = _torch()
        x = x.contiguous()
        if x.dtype != torch.float64 or x.ndim != 3 or x.shape[1] != self.n_cells:
            raise ValueError("x must be float64 [K, n_cells, t]")
        K, _, nt = (int(v) for v in x.shape)
        num = torch.empty((K, self.R, nt), dtype=torch.float64, device=x.device)
        den = torch.empty((self.R, nt), dtype=torch.float64, device=x.device)
        res = torch.empty((K, self.R, nt), dtype=torch.float64, device=x.device)
        _check(load().afhip_spatial_wavg(self._h, x.data_ptr(), K, nt, num.data_ptr(), den.data_ptr(),
                                         res.data_ptr(), _stream_ptr(x)))
        return num, den, res


class FusedPlan:
    """One pass over the raw cube for K output columns (afhip_plan_* in include/aggfly_hip.h).

    ``columns`` is a list of dicts: inner (calc name), inner_args (t0,t1,flag), transform
    (None | 'pow' | 'hinge'), transform_arg, outer (calc name or 'identity'), outer_args.
    """

    def __init__(self, T, n_cells, dtype_code, inner_bounds, outer_bounds, columns,
                 exact_order=False, tuning=0, device=None):
        """``device``: the GPU the plan's tables and scratch live on (default: the calling thread's current device); the
        cubes and the CSR it is run with must live there too (the library refuses others)."""
        lib = load()
        require_gpu()
        self.device_index = _device_index(device)
        self.ib = _i64(inner_bounds)
        self.ob = _i64(outer_bounds)
        self.K = len(columns)
        self.P = len(self.ob) - 1
        self.T, self.n_cells, self.dtype_code = int(T), int(n_cells), int(dtype_code)
        cols = (Column * self.K)()
        for j, c in enumerate(columns):
            cols[j].inner = CALC_CODE[c["inner"]]
            tf = c.get("transform")
            cols[j].transform = {None: TF_NONE, "pow": TF_POW, "hinge": TF_HINGE, "inter": TF_INTER}[tf]
            cols[j].transform_arg = float(c.get("transform_arg", 0.0))
            outer = c.get("outer", "identity")
            cols[j].outer = IDENTITY if outer == "identity" else CALC_CODE[outer]
            cols[j].rounding = int(c.get("rounding", 0))
            for i, v in enumerate(c.get("inner_args", (0.0, 0.0, 0.0))):
                cols[j].inner_args[i] = float(v)
            for i, v in enumerate(c.get("outer_args", (0.0, 0.0, 0.0))):
                cols[j].outer_args[i] = float(v)
        d = PlanDesc()
        d.T, d.n_cells, d.dtype, d.K = self.T, self.n_cells, self.dtype_code, self.K
        d.G1 = len(self.ib) - 1
        d.inner_bounds = self.ib.ctypes.data_as(C.POINTER(C.c_int64))
        d.P = self.P
        d.outer_bounds = self.ob.ctypes.data_as(C.POINTER(C.c_int64))
        d.columns = cols
        d.exact_order = 1 if exact_order else 0
        d.tuning = int(tuning)
        h = C.c_void_p()
        with _on_device(self.device_index):
            _check(lib.afhip_plan_create(C.byref(d), C.byref(h)))
        self._h = h
        self.G1 = len(self.ib) - 1
        self._inter = {}               # column -> the bound second cube (kept alive while bound)
        self._ws = None                # this plan's scratch: a block of torch's caching allocator, grown at need (`_workspace`)
        import threading
        #: held by a caller while it binds second cubes and enqueues a run: the handle owns scratch in HBM and must not be
        #: entered by two calls at once (include/aggfly_hip.h); `engine._run_fused_pass` takes it around bind + run
        self.lock = threading.RLock()

    def bind_inter(self, column: int, other):
        """Bind an 'inter' column's second cube: an HBM tensor [G1, n_cells...] (float32 / float64, time-major)."""
        other = other.contiguous()
        if not other.is_cuda or other.numel() != self.G1 * self.n_cells or other.shape[0] != self.G1:
            raise ValueError(f"inter array must be an HBM tensor of {self.G1} x {self.n_cells} elements (inner groups x cells)")
        _check(load().afhip_plan_bind_inter(self._h, int(column), other.data_ptr(), _dtype_code(other)))
        self._inter[int(column)] = other

    def describe(self) -> str:
        buf = C.create_string_buffer(2048)
        load().afhip_plan_describe(self._h, buf, 2048)
        return buf.value.decode()

    def workspace_bytes(self, csr: "CSR | None" = None) -> int:
        """Bytes of HBM scratch a run needs: of `run_temporal` (no ``csr``), or of a whole `run` against ``csr``."""
        if csr is None:
            return int(load().afhip_plan_workspace_bytes(self._h))
        return int(load().afhip_plan_run_workspace_bytes(self._h, csr.handle))

    def scratch_bytes(self) -> int:
        """HBM this plan pins while it is cached: its scratch block (once a run has allocated it; before, the temporal stage's
        share plus the cell-major panel, which is what a run will ask for up to the small sums area)."""
        if self._ws is not None:
            return int(self._ws.numel())
        return int(self.workspace_bytes()) + 8 * self.n_cells * (self.K + 1) * max(self.P, 1)

    def _workspace(self, nbytes: int, device):
        """The plan's own scratch as a CALLER-owned workspace of the library: a uint8 block from torch's caching allocator, kept
        on the plan and grown at need (the outgrown block goes back to the allocator; the stream-ordered allocator keeps it
        alive for kernels already enqueued on the current stream).  Where the library's own first hipMalloc lands decides
        7-9 % of the bin-count kernel on some boxes (profiles/r03_plan_order_probe.txt: the same plan is fast on a torch
        block), and a plan-owned block would be a second allocator beside torch's."""
        torch = _torch()
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=device)
        return self._ws

    def _check_cube(self, cube):
        cube, T, n_cells = _dev_cube(cube)
        if T != self.T or n_cells != self.n_cells or _dtype_code(cube) != self.dtype_code:
            raise ValueError(f"cube shape/dtype does not match the plan (T={self.T}, cells={self.n_cells})")
        return cube

    def run_temporal(self, cube):
        """-> cells[K, P, n_cells] float64 (the temporal stage's per-cell output)."""
        torch = _torch()
        cube = self._check_cube(cube)
        cells = torch.empty((self.K, self.P, self.n_cells), dtype=torch.float64, device=cube.device)
        ws = self._workspace(self.workspace_bytes(), cube.device)
        _check(load().afhip_plan_run_temporal(self._h, cube.data_ptr(), cells.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(cube)))
        return cells

    def run(self, cube, csr: CSR, want_cells=False, timed=False, out=None, workspace=None):
        """-> dict(num[K,R,P], den[R,P], res[K,R,P], cells?, kernel_ms?).  ``workspace``: an optional
        caller-owned uint8 HBM tensor of at least ``workspace_bytes(csr)``; None: the plan's own block of torch's caching
        allocator (`_workspace`); the string "library": the library's own hipMalloc'ed scratch, as a bare C caller gets it.

        ``want_cells=False`` (what `aggregate_dataset` asks for) lets the library skip the per-cell values: one gather over the
        period partials finishes the panel, and plans with several output periods reduce their cells by region inside the
        streaming kernel at every period end (``describe()`` names the route of the last run); ``want_cells=True`` and
        ``exact_order`` plans build the cell-major panel and sum in the table's order.  Same numbers to rounding either way."""
        torch = _torch()
        cube = self._check_cube(cube)
        dev = cube.device
        if out is None:
            out = {"num": torch.empty((self.K, csr.R, self.P), dtype=torch.float64, device=dev),
                   "den": torch.empty((csr.R, self.P), dtype=torch.float64, device=dev),
                   "res": torch.empty((self.K, csr.R, self.P), dtype=torch.float64, device=dev)}
            if want_cells:
                out["cells"] = torch.empty((self.K, self.P, self.n_cells), dtype=torch.float64, device=dev)
        ms = (C.c_float * 2)() if timed else None
        cells_ptr = out["cells"].data_ptr() if "cells" in out else None
        need = self.workspace_bytes(csr)
        if isinstance(workspace, str):
            if workspace != "library":
                raise ValueError("workspace must be None, an HBM tensor or 'library'")
            ws_ptr, ws_bytes = None, 0
        else:
            if workspace is None:
                workspace = self._workspace(need, dev)
            elif workspace.numel() * workspace.element_size() < need or not workspace.is_cuda:
                raise ValueError(f"workspace must be an HBM tensor of >= {need} bytes")
            ws_ptr, ws_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
        _check(load().afhip_plan_run(self._h, cube.data_ptr(), csr.handle, out["num"].data_ptr(),
                                     out["den"].data_ptr(), out["res"].data_ptr(), cells_ptr, ws_ptr, ws_bytes,
                                     _stream_ptr(cube), ms))
        if timed:
            out["kernel_ms"] = (float(ms[0]), float(ms[1]))
        return out

    def profile_begin(self, max_launches: int):
        """Arm HIP-event pairs around the temporal kernel of the next ``max_launches`` runs."""
        _check(load().afhip_plan_profile_begin(self._h, int(max_launches)))
        self._prof_cap = int(max_launches)

    def profile_end(self):
        """-> list of per-launch temporal-kernel durations in ms (waits for the events)."""
        cap = getattr(self, "_prof_cap", 0)
        buf = (C.c_float * max(cap, 1))()
        n = load().afhip_plan_profile_end(self._h, buf, cap)
        if n < 0:
            _check(E_HIP)
        return [float(buf[i]) for i in range(min(n, cap))]

    def close(self):
        if getattr(self, "_h", None):
            load().afhip_plan_destroy(self._h)
            self._h = None
        self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
