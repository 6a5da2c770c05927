"""ctypes binding of ``libaggfly_codec.so`` (aggfly_amd/csrc/blosc1.c): the host-side chunk codecs
of the ingestion path — Blosc-1 containers (blosclz / lz4 / lz4hc / zlib / zstd, byte- and
bit-shuffle), plain Zstandard frames, and a Blosc-LZ4 encoder for the writer side.

The reference decodes chunks through numcodecs inside its dask graph
(`aggfly/dataset/dataset.py:697-728`); here every chunk is decoded by native code on a host thread
(ctypes drops the GIL), or a whole slab of chunks at once on an OpenMP team.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaggfly_codec.so")
_lib = None

CODEC_NAMES = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}


class CodecError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CodecError(f"{LIB_PATH} is missing: build it with `make -C aggfly_amd/csrc codec` "
                             "or `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_PATH)
        lib.afcodec_last_error.restype = C.c_char_p
        lib.afcodec_have.argtypes = [C.c_int]
        lib.afcodec_blosc_info.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.afcodec_blosc_decode.restype = C.c_int64
        lib.afcodec_blosc_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        lib.afcodec_blosc_decode_mt.restype = C.c_int64
        lib.afcodec_blosc_decode_mt.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]
        lib.afcodec_blosc_decode_many.argtypes = [C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                                  C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64)]
        lib.afcodec_blosc_decode_files.argtypes = [C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                                   C.c_int, C.POINTER(C.c_int64)]
        lib.afcodec_decode_files.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                             C.c_int, C.POINTER(C.c_int64)]
        lib.afcodec_decode_ranges.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64)]
        lib.afcodec_blosc_bound.restype = C.c_int64
        lib.afcodec_blosc_bound.argtypes = [C.c_int64, C.c_int64]
        lib.afcodec_blosc_encode_lz4.restype = C.c_int64
        lib.afcodec_blosc_encode_lz4.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64]
        lib.afcodec_lz4_decode.restype = C.c_int64
        lib.afcodec_lz4_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        lib.afcodec_zstd_decode.restype = C.c_int64
        lib.afcodec_zstd_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        lib.afcodec_zstd_bound.restype = C.c_int64
        lib.afcodec_zstd_bound.argtypes = [C.c_int64]
        lib.afcodec_zstd_encode.restype = C.c_int64
        lib.afcodec_zstd_encode.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64]
        lib.afcodec_blosc_lz4_plan.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                               C.POINTER(C.c_int32), C.c_void_p]
        lib.afcodec_read_packed.argtypes = [C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_int64,
                                            C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


EXPORTS = ("afcodec_last_error", "afcodec_have", "afcodec_blosc_info", "afcodec_blosc_decode", "afcodec_blosc_decode_mt", "afcodec_blosc_decode_many",
           "afcodec_blosc_decode_files", "afcodec_decode_files", "afcodec_decode_ranges",
           "afcodec_blosc_bound", "afcodec_blosc_encode_lz4", "afcodec_zstd_decode", "afcodec_zstd_bound", "afcodec_zstd_encode", "afcodec_lz4_decode", "afcodec_blosc_lz4_plan", "afcodec_read_packed")


def _err(lib, what):
    return CodecError(f"{what}: {lib.afcodec_last_error().decode()}")


def _addr(buf):
    """Address of a bytes-like object without copying it."""
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data, buf.nbytes
    mv = memoryview(buf)
    if mv.readonly:
        return C.cast(C.c_char_p(bytes(buf) if not isinstance(buf, bytes) else buf), C.c_void_p).value, mv.nbytes
    return C.addressof(C.c_char.from_buffer(mv)), mv.nbytes


def blosc_info(buf) -> dict:
    lib = load()
    p, n = _addr(buf)
    nb, bs, ts, fl = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
    if lib.afcodec_blosc_info(p, n, C.byref(nb), C.byref(bs), C.byref(ts), C.byref(fl)):
        raise _err(lib, "blosc_info")
    f = fl.value
    return {"nbytes": nb.value, "blocksize": bs.value, "typesize": ts.value, "flags": f, "codec": CODEC_NAMES.get((f >> 5) & 7, "?"),
            "shuffle": 1 if f & 1 else (2 if f & 4 else 0), "stored": bool(f & 2), "split": not (f & 0x10)}


def blosc_decode(buf, out: np.ndarray | None = None, threads: int = 1) -> np.ndarray:
    """Decode one Blosc-1 chunk; ``out`` (C-contiguous, >= nbytes) is filled in place when given;
    ``threads`` > 1 spreads the chunk's blocks over an OpenMP team (for large chunks)."""
    lib = load()
    p, n = _addr(buf)
    if out is None:
        out = np.empty(blosc_info(buf)["nbytes"], dtype=np.uint8)
    if not out.flags.c_contiguous:
        raise ValueError("blosc_decode: out must be C-contiguous")
    if threads > 1:
        r = lib.afcodec_blosc_decode_mt(p, n, out.ctypes.data, out.nbytes, int(threads))
    else:
        r = lib.afcodec_blosc_decode(p, n, out.ctypes.data, out.nbytes)
    if r < 0:
        raise _err(lib, "blosc_decode")
    return out


def blosc_decode_many(bufs, outs, threads: int = 8):
    """Decode chunks ``bufs[i]`` into the C-contiguous arrays ``outs[i]`` on an OpenMP team."""
    lib = load()
    n = len(bufs)
    keep = [b if isinstance(b, (bytes, np.ndarray)) else bytes(b) for b in bufs]
    addrs = [_addr(b) for b in keep]
    cp = (C.c_void_p * n)(*[a for a, _ in addrs])
    cs = (C.c_int64 * n)(*[s for _, s in addrs])
    dp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    ds = (C.c_int64 * n)(*[o.nbytes for o in outs])
    res = (C.c_int64 * n)()
    if lib.afcodec_blosc_decode_many(n, cp, cs, dp, ds, int(threads), res):
        bad = [i for i in range(n) if res[i] < 0]
        raise CodecError(f"blosc_decode_many: chunks {bad[:8]} failed: {lib.afcodec_last_error().decode()}")
    return [int(res[i]) for i in range(n)]


KIND = {"raw": 0, "blosc": 1, "zstd": 2, "zlib": 3, "gzip": 3, "lz4": 4}


def _require_full(what, names, outs, res):
    """A present chunk must fill its destination exactly: a Blosc header announcing fewer bytes, a truncated raw file or
    a zstd / zlib frame that decodes short would otherwise leave the rest of a (re-used, page-locked) staging slot to
    travel to HBM as data.  zarr / numcodecs raise on such chunks (`aggfly/dataset/dataset.py:697-728` reads through them)."""
    short = [(names[i], int(res[i]), outs[i].nbytes) for i in range(len(outs)) if res[i] != -100 and int(res[i]) != outs[i].nbytes]
    if short:
        nm, got, want = short[0]
        raise CodecError(f"{what}: chunk {nm} decoded to {got} bytes, expected {want}"
                         + (f" (and {len(short) - 1} more)" if len(short) > 1 else ""))


def decode_files(kind: str, paths, outs, threads: int = 8, exact: bool = True):
    """Read and decode the chunk files ``paths[i]`` (all of codec ``kind``: raw / blosc / zstd / zlib /
    gzip) into ``outs[i]`` on an OpenMP team.  -> list of decoded sizes; -100 marks a missing file (the
    caller applies the fill value).  ``exact``: a present chunk must decode to exactly ``outs[i].nbytes``."""
    lib = load()
    n = len(paths)
    pp = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    dp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    ds = (C.c_int64 * n)(*[o.nbytes for o in outs])
    res = (C.c_int64 * n)()
    if lib.afcodec_decode_files(KIND[kind], n, pp, dp, ds, int(threads), res):
        bad = [paths[i] for i in range(n) if res[i] < 0 and res[i] != -100]
        raise CodecError(f"decode_files({kind}): {bad[:4]} failed: {lib.afcodec_last_error().decode()}")
    if exact:
        _require_full(f"decode_files({kind})", paths, outs, res)
    return [int(res[i]) for i in range(n)]


def decode_ranges(kind: str, locators, outs, threads: int = 8, exact: bool = True):
    """Like `decode_files` for ``locators[i] = (path, offset, nbytes)`` — nbytes < 0 = the whole file, None =
    an absent chunk (reported as -100 without touching the disk).  ``kind`` may be a pair ``(codec, element_size)``:
    the codec's output is then byte-unshuffled (HDF5 shuffle + deflate chunks).  ``exact`` as in `decode_files`."""
    lib = load()
    n = len(locators)
    pp = (C.c_char_p * n)(*[os.fsencode(l[0]) if l is not None else b"" for l in locators])
    offs = (C.c_int64 * n)(*[int(l[1]) if l is not None else 0 for l in locators])
    lens = (C.c_int64 * n)(*[int(l[2]) if l is not None else -1 for l in locators])
    dp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    ds = (C.c_int64 * n)(*[o.nbytes for o in outs])
    res = (C.c_int64 * n)()
    code = KIND[kind] if isinstance(kind, str) else KIND[kind[0]] + 16 * int(kind[1])      # (codec, element size): + byte unshuffle
    if lib.afcodec_decode_ranges(code, n, pp, offs, lens, dp, ds, int(threads), res):
        bad = [locators[i] for i in range(n) if res[i] < 0 and res[i] != -100]
        raise CodecError(f"decode_ranges({kind}): {bad[:4]} failed: {lib.afcodec_last_error().decode()}")
    if exact:
        _require_full(f"decode_ranges({kind})", locators, outs, res)
    return [int(res[i]) for i in range(n)]


def read_packed(locators, dst: np.ndarray, align: int = 64, threads: int = 8):
    """`afcodec_read_packed`: the byte ranges ``locators[i] = (path, offset, nbytes)`` (nbytes < 0: the whole file; None: an
    absent chunk) read back to back into the uint8 array ``dst`` -> (offsets int64[n + 1], sizes int64[n], -100 = missing).
    The files are neither stat'ed nor opened from Python."""
    lib = load()
    n = len(locators)
    pp = (C.c_char_p * n)(*[os.fsencode(l[0]) if l is not None else b"" for l in locators])
    offs = (C.c_int64 * n)(*[int(l[1]) if l is not None else 0 for l in locators])
    lens = (C.c_int64 * n)(*[int(l[2]) if l is not None else -1 for l in locators])
    out_off = np.zeros(n + 1, dtype=np.int64)
    res = np.zeros(n, dtype=np.int64)
    if lib.afcodec_read_packed(n, pp, offs, lens, dst.ctypes.data, dst.nbytes, int(align), int(threads), out_off.ctypes.data, res.ctypes.data):
        bad = [locators[i] for i in range(n) if res[i] < 0 and res[i] != -100]
        raise CodecError(f"read_packed: {bad[:4]} failed: {lib.afcodec_last_error().decode()}")
    return out_off, res


def blosc_decode_files(paths, outs, threads: int = 8):
    return decode_files("blosc", paths, outs, threads)


def blosc_encode(data, typesize: int, shuffle: bool = True, blocksize: int = 0) -> bytes:
    """Blosc-1 / LZ4 chunk of ``data`` (bytes-like or array): what numcodecs' ``Blosc(cname="lz4")`` reads."""
    lib = load()
    arr = np.ascontiguousarray(data) if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    cap = lib.afcodec_blosc_bound(arr.nbytes, blocksize)
    dst = np.empty(cap, dtype=np.uint8)
    r = lib.afcodec_blosc_encode_lz4(arr.ctypes.data, arr.nbytes, int(typesize), 1 if shuffle else 0, int(blocksize), dst.ctypes.data, cap)
    if r < 0:
        raise _err(lib, "blosc_encode")
    return dst[:r].tobytes()


def zstd_decode(buf, nbytes: int, out: np.ndarray | None = None) -> np.ndarray:
    """Decode one Zstandard frame of at most ``nbytes`` decoded bytes."""
    lib = load()
    p, n = _addr(buf)
    if out is None:
        out = np.empty(nbytes, dtype=np.uint8)
    r = lib.afcodec_zstd_decode(p, n, out.ctypes.data, out.nbytes)
    if r < 0:
        raise _err(lib, "zstd_decode")
    return out[:r] if r != out.nbytes and out.ndim == 1 else out


def zstd_encode(data, level: int = 3) -> bytes:
    lib = load()
    arr = np.ascontiguousarray(data) if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    cap = lib.afcodec_zstd_bound(arr.nbytes)
    dst = np.empty(cap, dtype=np.uint8)
    r = lib.afcodec_zstd_encode(arr.ctypes.data, arr.nbytes, int(level), dst.ctypes.data, cap)
    if r < 0:
        raise _err(lib, "zstd_encode")
    return dst[:r].tobytes()


def lz4_decode(buf, nbytes: int, out: np.ndarray | None = None) -> np.ndarray:
    """numcodecs LZ4 chunk (int32 size header + raw block) of at most ``nbytes`` decoded bytes."""
    lib = load()
    p, n = _addr(buf)
    if out is None:
        out = np.empty(nbytes, dtype=np.uint8)
    r = lib.afcodec_lz4_decode(p, n, out.ctypes.data, out.nbytes)
    if r < 0:
        raise _err(lib, "lz4_decode")
    return out[:r] if r != out.nbytes and out.ndim == 1 else out


# record layouts shared with libaggfly_hip (include/aggfly_hip.h: afhip_lz4_stream, afhip_shuffle_block)
LZ4_STREAM = np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("csize", "<i4"), ("dsize", "<i4"), ("to_out", "<i4"), ("pad", "<i4")])
SHUFFLE_BLOCK = np.dtype([("tmp_off", "<i8"), ("out_off", "<i8"), ("bsize", "<i4"), ("typesize", "<i4")])
E_UNSUPPORTED = -2


class PlanCapacityError(CodecError):
    """`blosc_lz4_plan`: the chunks need more stream / block records than the lists passed in hold (their geometry differs
    from the one the lists were sized for)."""


def blosc_lz4_plan(base: np.ndarray, comp_off, comp_size, out_off, out_size, streams: np.ndarray, blocks: np.ndarray):
    """Plan the GPU-side decode of Blosc-1 chunks that sit in ``base`` (uint8; chunk i = ``comp_size[i]`` bytes at
    ``comp_off[i]``): fills ``streams`` (dtype `LZ4_STREAM`) and ``blocks`` (dtype `SHUFFLE_BLOCK`) for
    `hip.lz4_decode_streams` / `hip.unshuffle_blocks`.  -> (n_streams, n_blocks, tmp_bytes, max_dsize, results);
    ``results[i]`` = decoded size, or `E_UNSUPPORTED` for a chunk the GPU route does not take (decode it on the host);
    malformed containers raise `CodecError`."""
    lib = load()
    n = len(comp_off)
    co, cs, oo, osz = (np.ascontiguousarray(a, dtype=np.int64) for a in (comp_off, comp_size, out_off, out_size))
    res = np.zeros(n, dtype=np.int64)
    ns, nb, tmp, maxd = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int32(0)
    assert streams.dtype == LZ4_STREAM and blocks.dtype == SHUFFLE_BLOCK and base.dtype == np.uint8
    rc = lib.afcodec_blosc_lz4_plan(base.ctypes.data, n, co.ctypes.data, cs.ctypes.data, oo.ctypes.data, osz.ctypes.data,
                                    streams.ctypes.data, len(streams), C.byref(ns), blocks.ctypes.data, len(blocks), C.byref(nb),
                                    C.byref(tmp), C.byref(maxd), res.ctypes.data)
    bad = [int(i) for i in np.nonzero((res < 0) & (res != E_UNSUPPORTED))[0]]
    if bad or (rc and rc != E_UNSUPPORTED):
        msg = lib.afcodec_last_error().decode()
        if not bad and "list too small" in msg:       # more streams / blocks than the caller's record lists hold: not damage
            raise PlanCapacityError(f"blosc_lz4_plan: {msg}")
        raise CodecError(f"blosc_lz4_plan: chunks {bad[:8]} are malformed: {msg}")
    return int(ns.value), int(nb.value), int(tmp.value), int(maxd.value), res
