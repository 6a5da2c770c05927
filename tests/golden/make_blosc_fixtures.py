"""Generates tests/golden/blosc_fixtures.json: chunks compressed by the REAL c-blosc library
(/opt/conda/lib/libblosc.so.1, v1.21 — present in the build container only) for the in-tree
Blosc-1 decoder (aggfly_amd/csrc/blosc1.c) to be checked against.  Inputs are recipes (seeded
numpy), so only the compressed bytes and a SHA-256 of the raw bytes are stored.

    python tests/golden/make_blosc_fixtures.py        # here, not on the GPU box
"""
import base64
import ctypes as C
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = "/opt/conda/lib/libblosc.so.1"


def recipe(name, n, dtype, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype(dtype)
    if name == "temperature":          # smooth field + noise: what climate chunks look like
        x = 280 + 15 * np.sin(np.arange(n) / 37.0) + rng.normal(0, 0.5, n)
    elif name == "random":             # incompressible: streams stored raw
        x = rng.normal(0, 1e6, n)
    elif name == "constant":           # long runs
        x = np.full(n, 273.15)
    elif name == "steps":
        x = np.repeat(rng.integers(-50, 50, n // 16 + 1), 16)[:n].astype(float)
    elif name == "nanmask":
        x = 10 + rng.normal(0, 3, n)
        x[rng.random(n) < 0.3] = np.nan
    else:
        raise KeyError(name)
    if dt.kind in "iu":
        return np.round(x).astype(dt)
    return x.astype(dt)


CASES = []
for cname in ("lz4", "lz4hc", "zstd", "zlib", "blosclz"):
    for shuffle in (0, 1, 2):
        for dtype, n in (("<f4", 6000), ("<f8", 3001), ("<i2", 5000)):
            CASES.append(dict(cname=cname, shuffle=shuffle, dtype=dtype, n=n, recipe="temperature", clevel=5, blocksize=0))
    CASES.append(dict(cname=cname, shuffle=1, dtype="<f4", n=40000, recipe="temperature", clevel=5, blocksize=16384))   # several blocks + leftover
    CASES.append(dict(cname=cname, shuffle=1, dtype="<f4", n=9000, recipe="random", clevel=5, blocksize=8192))
    CASES.append(dict(cname=cname, shuffle=1, dtype="<f8", n=5000, recipe="constant", clevel=9, blocksize=0))
    CASES.append(dict(cname=cname, shuffle=1, dtype="<f4", n=7777, recipe="nanmask", clevel=1, blocksize=4096))
    CASES.append(dict(cname=cname, shuffle=2, dtype="<f4", n=8191, recipe="steps", clevel=5, blocksize=4096))
    CASES.append(dict(cname=cname, shuffle=1, dtype="<f4", n=20, recipe="temperature", clevel=5, blocksize=0))            # tiny: memcpy'd
    CASES.append(dict(cname=cname, shuffle=1, dtype="<f4", n=5000, recipe="temperature", clevel=0, blocksize=0))          # clevel 0: memcpy'd


def main():
    lib = C.CDLL(LIB)
    lib.blosc_compress_ctx.restype = C.c_int
    lib.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_char_p, C.c_size_t, C.c_int]
    lib.blosc_decompress_ctx.restype = C.c_int
    lib.blosc_get_version_string.restype = C.c_char_p
    out = {"generator": f"c-blosc {lib.blosc_get_version_string().decode()} ({LIB})", "cases": []}
    for i, c in enumerate(CASES):
        raw = recipe(c["recipe"], c["n"], c["dtype"], seed=1000 + i)
        src = raw.tobytes()
        dst = C.create_string_buffer(len(src) + 16 + 4096)
        nb = lib.blosc_compress_ctx(c["clevel"], c["shuffle"], raw.dtype.itemsize, len(src), src, dst, len(dst),
                                    c["cname"].encode(), c["blocksize"], 1)
        assert nb > 0, (c, nb)
        back = C.create_string_buffer(len(src))
        assert lib.blosc_decompress_ctx(dst, back, len(src), 1) == len(src) and back.raw == src
        out["cases"].append(dict(c, seed=1000 + i, sha256=hashlib.sha256(src).hexdigest(), nbytes=len(src), cbytes=nb,
                                 chunk_b64=base64.b64encode(dst.raw[:nb]).decode()))
    with open(os.path.join(HERE, "blosc_fixtures.json"), "w") as f:
        json.dump(out, f)
    print(len(out["cases"]), "cases,", sum(c["cbytes"] for c in out["cases"]), "compressed bytes,", out["generator"])


if __name__ == "__main__":
    main()
