"""Host-side chunk codecs of the ingestion path (aggfly_amd/csrc/blosc1.c) against chunks written by
the REAL c-blosc 1.21 (tests/golden/blosc_fixtures.json, made by tests/golden/make_blosc_fixtures.py),
and the Zarr reader on stores built from those chunks.  CPU only."""
import base64
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_blosc_fixtures import recipe          # noqa: E402  (seeded numpy recipes of the raw arrays)

from aggfly_amd import codec, io as afio        # noqa: E402

FIX = json.load(open(os.path.join(HERE, "golden", "blosc_fixtures.json")))
CASES = FIX["cases"]


def _id(c):
    return f"{c['cname']}-s{c['shuffle']}-{c['dtype'][1:]}-{c['recipe']}-{c['n']}-b{c['blocksize']}-l{c['clevel']}"


def test_library_exports_every_symbol():
    import re
    lib = codec.load()
    hdr = open(os.path.join(os.path.dirname(HERE), "include", "aggfly_codec.h")).read()
    declared = set(re.findall(r"\b(afcodec_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(codec.EXPORTS), declared ^ set(codec.EXPORTS)
    assert all(hasattr(lib, s) for s in declared)
    assert lib.afcodec_have(1) and lib.afcodec_have(4) and lib.afcodec_have(3) and lib.afcodec_have(0)


@pytest.mark.parametrize("case", CASES, ids=_id)
def test_decodes_real_cblosc_chunks_bit_exact(case):
    chunk = base64.b64decode(case["chunk_b64"])
    raw = recipe(case["recipe"], case["n"], case["dtype"], case["seed"])
    assert hashlib.sha256(raw.tobytes()).hexdigest() == case["sha256"]          # the recipe still yields the fixture's input
    info = codec.blosc_info(chunk)
    assert info["nbytes"] == raw.nbytes and info["typesize"] == raw.dtype.itemsize
    assert info["codec"] == {"lz4hc": "lz4"}.get(case["cname"], case["cname"]) or info["stored"]
    out = codec.blosc_decode(chunk)
    assert out.tobytes() == raw.tobytes()
    into = np.empty(raw.shape, dtype=raw.dtype)                                 # straight into a typed destination
    codec.blosc_decode(chunk, into)
    assert into.tobytes() == raw.tobytes()
    into[...] = 0
    codec.blosc_decode(chunk, into, threads=3)                                  # blocks spread over a team
    assert into.tobytes() == raw.tobytes()


def test_decode_many_on_threads():
    chunks = [base64.b64decode(c["chunk_b64"]) for c in CASES]
    outs = [np.empty(c["nbytes"], dtype=np.uint8) for c in CASES]
    res = codec.blosc_decode_many(chunks, outs, threads=4)
    assert res == [c["nbytes"] for c in CASES]
    assert all(hashlib.sha256(o.tobytes()).hexdigest() == c["sha256"] for o, c in zip(outs, CASES))


def test_damaged_chunks_are_refused_not_crashed_on():
    chunk = base64.b64decode(next(c for c in CASES if c["cname"] == "lz4" and c["n"] == 40000)["chunk_b64"])
    with pytest.raises(codec.CodecError):
        codec.blosc_decode(chunk[:10])                                          # shorter than the header
    with pytest.raises(codec.CodecError):
        codec.blosc_decode(chunk[:len(chunk) // 2])                             # truncated
    bad = bytearray(chunk); bad[0] = 9
    with pytest.raises(codec.CodecError):
        codec.blosc_decode(bytes(bad))                                          # unknown format version
    bad = bytearray(chunk); bad[16:20] = (2 ** 31 - 1).to_bytes(4, "little")
    with pytest.raises(codec.CodecError):
        codec.blosc_decode(bytes(bad))                                          # block offset out of range
    bad = bytearray(chunk); bad[len(bad) // 2:len(bad) // 2 + 64] = os.urandom(64)
    try:
        out = codec.blosc_decode(bytes(bad))                                    # garbage payload: error or wrong bytes, never a fault
        assert out.nbytes == 160000
    except codec.CodecError:
        pass
    with pytest.raises(codec.CodecError):
        codec.blosc_decode(chunk, np.empty(100, dtype=np.uint8))                # destination too small


@pytest.mark.parametrize("dtype,n,shuffle,blocksize", [("<f4", 100000, True, 0), ("<f8", 33333, True, 65536), ("<f4", 17, True, 0),
                                                        ("<i2", 5000, False, 4096), ("<f4", 70001, True, 10000), ("<f4", 0, True, 0)])
def test_encoder_round_trip_and_real_cblosc_reads_it(dtype, n, shuffle, blocksize):
    rng = np.random.default_rng(n)
    x = (280 + 10 * np.sin(np.arange(n) / 50) + rng.normal(0, 0.3, n)).astype(dtype)
    enc = codec.blosc_encode(x, x.dtype.itemsize, shuffle, blocksize)
    assert codec.blosc_decode(enc).tobytes() == x.tobytes()
    if n > 1000:
        assert len(enc) < x.nbytes                                              # it does compress
    real = "/opt/conda/lib/libblosc.so.1"                                       # present in the build container only
    if os.path.exists(real) and n:
        lib = C.CDLL(real)
        back = C.create_string_buffer(x.nbytes)
        assert lib.blosc_decompress_ctx(enc, back, x.nbytes, 1) == x.nbytes and back.raw == x.tobytes()


def _zarr_v2_from_chunks(path, name, shape, chunks, dtype, compressor, files, dims):
    d = os.path.join(path, name)
    os.makedirs(d, exist_ok=True)
    json.dump({"zarr_format": 2, "shape": list(shape), "chunks": list(chunks), "dtype": dtype, "compressor": compressor,
               "fill_value": "NaN", "order": "C", "filters": None}, open(os.path.join(d, ".zarray"), "w"))
    json.dump({"_ARRAY_DIMENSIONS": list(dims)}, open(os.path.join(d, ".zattrs"), "w"))
    for key, blob in files.items():
        open(os.path.join(d, key), "wb").write(blob)


@pytest.mark.parametrize("cname,shuffle", [("lz4", 1), ("zstd", 1), ("blosclz", 1), ("zlib", 2), ("lz4hc", 0)])
def test_zarr_store_made_of_real_cblosc_chunks(tmp_path, cname, shuffle):
    """A Zarr v2 array whose chunk files are byte-for-byte what numcodecs' Blosc would have written."""
    case = next(c for c in CASES if c["cname"] == cname and c["shuffle"] == shuffle and c["dtype"] == "<f4" and c["n"] == 6000)
    raw = recipe(case["recipe"], case["n"], case["dtype"], case["seed"]).reshape(60, 10, 10)
    comp = {"id": "blosc", "cname": cname, "clevel": 5, "shuffle": shuffle, "blocksize": 0}
    _zarr_v2_from_chunks(str(tmp_path), "t2m", (60, 10, 10), (60, 10, 10), "<f4", comp,
                         {"0.0.0": base64.b64decode(case["chunk_b64"])}, ("time", "latitude", "longitude"))
    got = afio.ZarrArray(os.path.join(str(tmp_path), "t2m")).read()
    np.testing.assert_array_equal(got, raw)


def test_zarr_zstd_compressor_and_blosc_writer(tmp_path):
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(2)
    data = rng.normal(10, 5, (48, 6, 8)).astype("<f8")
    files = {}
    for it in range(2):
        files[f"{it}.0.0"] = pa.Codec("zstd").compress(data[it * 24:(it + 1) * 24].tobytes(), asbytes=True)
    _zarr_v2_from_chunks(str(tmp_path), "v", data.shape, (24, 6, 8), "<f8", {"id": "zstd", "level": 3}, files,
                         ("time", "latitude", "longitude"))
    np.testing.assert_array_equal(afio.ZarrArray(os.path.join(str(tmp_path), "v")).read(threads=2), data)
    # numcodecs' LZ4 codec: int32 decoded size + one raw LZ4 block
    files = {f"{it}.0.0": len(data[it * 24:(it + 1) * 24].tobytes()).to_bytes(4, "little")
             + pa.Codec("lz4_raw").compress(data[it * 24:(it + 1) * 24].tobytes(), asbytes=True) for it in range(2)}
    _zarr_v2_from_chunks(str(tmp_path), "l", data.shape, (24, 6, 8), "<f8", {"id": "lz4", "acceleration": 1}, files,
                         ("time", "latitude", "longitude"))
    za = afio.ZarrArray(os.path.join(str(tmp_path), "l"))
    assert za.native_kind == "lz4"
    np.testing.assert_array_equal(za.read(threads=2), data)
    outs = [np.empty((24, 6, 8)), np.empty((24, 6, 8))]
    assert codec.decode_ranges("lz4", [za.chunk_locator((0, 0, 0)), za.chunk_locator((1, 0, 0))], outs, threads=2) == [9216, 9216]
    np.testing.assert_array_equal(np.concatenate(outs), data)
    # numcodecs' Shuffle filter in front of a zlib compressor (what HDF5-style shuffle + deflate stores look like)
    import zlib as _zl
    files = {}
    for it in range(2):
        raw = np.frombuffer(data[it * 24:(it + 1) * 24].tobytes(), dtype=np.uint8)
        files[f"{it}.0.0"] = _zl.compress(raw.reshape(-1, 8).T.tobytes(), 1)           # byte planes, then deflate
    _zarr_v2_from_chunks(str(tmp_path), "f", data.shape, (24, 6, 8), "<f8", {"id": "zlib", "level": 1}, files,
                         ("time", "latitude", "longitude"))
    meta = json.load(open(os.path.join(str(tmp_path), "f", ".zarray")))
    meta["filters"] = [{"id": "shuffle", "elementsize": 8}]
    json.dump(meta, open(os.path.join(str(tmp_path), "f", ".zarray"), "w"))
    zf = afio.ZarrArray(os.path.join(str(tmp_path), "f"))
    assert zf.native_kind is None                                                # a chain: host route
    np.testing.assert_array_equal(zf.read(), data)
    meta["filters"] = [{"id": "delta", "dtype": "<f8"}]
    json.dump(meta, open(os.path.join(str(tmp_path), "f", ".zarray"), "w"))
    with pytest.raises(ValueError):
        afio.ZarrArray(os.path.join(str(tmp_path), "f"))
    # the writer's default is Blosc-LZ4 + shuffle, ragged edge chunks included
    afio._write_array(str(tmp_path), "w", data, ("time", "latitude", "longitude"), (20, 4, 8), {},
                      {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0})
    za = afio.ZarrArray(os.path.join(str(tmp_path), "w"))
    assert codec.blosc_info(open(os.path.join(za.path, "2.1.0"), "rb").read())["codec"] == "lz4"
    np.testing.assert_array_equal(za.read(), data)


@pytest.mark.parametrize("compress", ["zstd", "blosc", "zlib", False])
def test_zarr_format_3_store_round_trip(tmp_path, compress):
    """Format 3 (zarr.json, c/ chunk keys, codec pipeline) as zarr-python 3 lays it out; written by this
    package's own writer (no zarr-python 3 exists in the image to produce an independent fixture)."""
    import pandas as pd
    import aggfly_amd as af
    rng = np.random.default_rng(4)
    T, ny, nx = 50, 7, 9
    cube = rng.normal(280, 8, (T, ny, nx)).astype(np.float32)
    cube[3, 2, 1] = np.nan
    time = pd.date_range("2003-02-01", periods=T, freq="h")
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                                 {"time": time, "latitude": 10 + np.arange(ny) * 0.5, "longitude": 100 + np.arange(nx) * 0.5}), lon_is_360=True)
    store = str(tmp_path / "v3.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 16, "latitude": 4, "longitude": 9}, compress=compress, zarr_format=3)
    assert os.path.exists(os.path.join(store, "zarr.json")) and os.path.exists(os.path.join(store, "t2m", "c", "3", "1", "0"))
    meta = json.load(open(os.path.join(store, "t2m", "zarr.json")))
    assert meta["zarr_format"] == 3 and meta["node_type"] == "array" and meta["dimension_names"] == ["time", "latitude", "longitude"]
    assert [c["name"] for c in meta["codecs"]] == ["bytes"] + ({"zstd": ["zstd"], "blosc": ["blosc"], "zlib": ["gzip"], False: []}[compress])
    back = af.dataset_from_path(store, "t2m")
    np.testing.assert_array_equal(back.cube(), cube)
    assert back.time.equals(time) and np.array_equal(back.latitude, ds.latitude)


def test_zarr_format_3_details(tmp_path):
    """v2-style chunk keys inside a format-3 array, a crc32c trailer, big-endian bytes, a missing chunk
    (fill value), and the refusals."""
    d = str(tmp_path / "a")
    os.makedirs(d)
    data = np.arange(24, dtype=">i4").reshape(4, 6)
    meta = {"zarr_format": 3, "node_type": "array", "shape": [4, 6], "data_type": "int32",
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [2, 6]}},
            "chunk_key_encoding": {"name": "v2", "configuration": {"separator": "."}}, "fill_value": -1,
            "codecs": [{"name": "bytes", "configuration": {"endian": "big"}}, {"name": "gzip", "configuration": {"level": 1}}, {"name": "crc32c"}],
            "attributes": {"units": "1"}, "dimension_names": ["y", "x"]}
    json.dump(meta, open(os.path.join(d, "zarr.json"), "w"))
    import gzip as _g
    open(os.path.join(d, "0.0"), "wb").write(_g.compress(data[:2].tobytes()) + bytes(4))
    za = afio.ZarrArray(d)
    got = za.read()
    assert za.dims == ("y", "x") and got.dtype == np.dtype("int32")
    np.testing.assert_array_equal(got[:2], np.arange(12).reshape(2, 6))
    assert (got[2:] == -1).all()                                          # chunk 1.0 is absent
    for bad in ({"codecs": [{"name": "sharding_indexed", "configuration": {}}]}, {"data_type": "complex64"},
                {"codecs": [{"name": "bytes"}, {"name": "sharding_indexed", "configuration": {"chunk_shape": [1, 6]}}]},
                {"chunk_grid": {"name": "rectilinear", "configuration": {}}}):
        json.dump(dict(meta, **bad), open(os.path.join(d, "zarr.json"), "w"))
        with pytest.raises(ValueError):
            afio.ZarrArray(d)


@pytest.mark.parametrize("compress", ["zstd", "blosc", False])
def test_zarr_format_3_sharded_store(tmp_path, compress):
    """`sharding_indexed`: several chunks bundled per shard file, located through the (offset, nbytes) index at
    the end of the shard (crc32c trailer); ragged edges, an inner chunk outside the array (empty entry) and a
    missing shard (fill value).  Written by this package's writer — no zarr-python 3 here to cross-check."""
    import pandas as pd
    import aggfly_amd as af
    rng = np.random.default_rng(9)
    T, ny, nx = 50, 7, 9
    cube = rng.normal(280, 8, (T, ny, nx)).astype(np.float32)
    time = pd.date_range("2003-02-01", periods=T, freq="h")
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                                 {"time": time, "latitude": 10 + np.arange(ny) * 0.5, "longitude": 100 + np.arange(nx) * 0.5}), lon_is_360=True)
    store = str(tmp_path / "sh.zarr")
    af.dataset_to_zarr(ds, store, var="t2m", chunks={"time": 8, "latitude": 4, "longitude": 3}, shards={"time": 24, "latitude": 8, "longitude": 9},
                       compress=compress, zarr_format=3)
    meta = json.load(open(os.path.join(store, "t2m", "zarr.json")))
    assert meta["codecs"][0]["name"] == "sharding_indexed" and meta["chunk_grid"]["configuration"]["chunk_shape"] == [24, 8, 9]
    assert sorted(os.listdir(os.path.join(store, "t2m", "c"))) == ["0", "1", "2"]          # 3 shards along time, 1 x 1 in space
    za = afio.ZarrArray(os.path.join(store, "t2m"))
    assert za.chunks == (8, 4, 3) and za.shard_shape == (24, 8, 9)
    assert za.chunk_locator((0, 1, 2)) is not None and za.chunk_locator((6, 1, 2)) is not None
    np.testing.assert_array_equal(za.read(), cube)
    back = af.dataset_from_path(store, "t2m")
    np.testing.assert_array_equal(back.cube(), cube)
    os.remove(os.path.join(store, "t2m", "c", "1", "0", "0"))                               # a missing shard = fill value
    got = afio.ZarrArray(os.path.join(store, "t2m")).read()
    assert np.isnan(got[24:48]).all()
    np.testing.assert_array_equal(got[:24], cube[:24])


def test_short_chunks_are_refused(tmp_path):
    """A present chunk must decode to exactly the chunk size: a truncated raw file, a zstd frame of fewer bytes or a Blosc
    chunk whose header announces less than the array's chunk would otherwise leave stale staging bytes to travel on as
    data (zarr / numcodecs raise on such chunks too)."""
    import pyarrow as pa
    rng = np.random.default_rng(9)
    data = rng.normal(size=(48, 6, 8))
    cb = 24 * 6 * 8 * 8
    # raw store, second chunk file cut short
    files = {"0.0.0": data[:24].tobytes(), "1.0.0": data[24:].tobytes()[:cb - 64]}
    _zarr_v2_from_chunks(str(tmp_path), "r", data.shape, (24, 6, 8), "<f8", None, files, ("time", "latitude", "longitude"))
    za = afio.ZarrArray(os.path.join(str(tmp_path), "r"))
    assert za.native_kind == "raw"
    outs = [np.empty((24, 6, 8)), np.empty((24, 6, 8))]
    locs = [za.chunk_locator((0, 0, 0)), za.chunk_locator((1, 0, 0))]
    with pytest.raises(codec.CodecError, match=r"1\.0\.0.*decoded to 9152 bytes, expected 9216"):
        codec.decode_ranges("raw", locs, outs, threads=2)
    assert codec.decode_ranges("raw", locs, outs, threads=2, exact=False) == [9216, 9152]      # the old, lenient answer
    # zstd frame holding 23 of the 24 steps
    files = {"0.0.0": pa.Codec("zstd").compress(data[:24].tobytes(), asbytes=True),
             "1.0.0": pa.Codec("zstd").compress(data[24:47].tobytes(), asbytes=True)}
    _zarr_v2_from_chunks(str(tmp_path), "z", data.shape, (24, 6, 8), "<f8", {"id": "zstd", "level": 1}, files,
                         ("time", "latitude", "longitude"))
    zz = afio.ZarrArray(os.path.join(str(tmp_path), "z"))
    with pytest.raises(codec.CodecError, match="decoded to 8832 bytes, expected 9216"):
        codec.decode_ranges("zstd", [zz.chunk_locator((0, 0, 0)), zz.chunk_locator((1, 0, 0))], outs, threads=2)
    # Blosc chunk (written by the in-tree encoder) of fewer elements than the array's chunk
    files = {"0.0.0": codec.blosc_encode(data[:24].tobytes(), 8), "1.0.0": codec.blosc_encode(data[24:40].tobytes(), 8)}
    _zarr_v2_from_chunks(str(tmp_path), "b", data.shape, (24, 6, 8), "<f8",
                         {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}, files, ("time", "latitude", "longitude"))
    zb = afio.ZarrArray(os.path.join(str(tmp_path), "b"))
    with pytest.raises(codec.CodecError, match="expected 9216"):
        codec.decode_files("blosc", [zb.chunk_locator((i, 0, 0))[0] for i in range(2)], outs, threads=2)
    # an absent chunk is not a short chunk
    os.remove(za.chunk_locator((1, 0, 0))[0])
    assert codec.decode_ranges("raw", [za.chunk_locator((0, 0, 0)), za.chunk_locator((1, 0, 0))], outs, threads=2) == [9216, -100]


def _plan_one(chunk: bytes, nbytes: int):
    base = np.frombuffer(chunk, dtype=np.uint8).copy()
    streams = np.zeros(4096, dtype=codec.LZ4_STREAM)
    blocks = np.zeros(1024, dtype=codec.SHUFFLE_BLOCK)
    ns, nb, tmpb, maxd, res = codec.blosc_lz4_plan(base, [0], [len(chunk)], [0], [nbytes], streams, blocks)
    return base, streams[:ns], blocks[:nb], tmpb, maxd, int(res[0])


@pytest.mark.parametrize("case", CASES, ids=_id)
def test_gpu_decode_plan_on_real_cblosc_chunks(case):
    """`afcodec_blosc_lz4_plan` (the host half of the GPU-side decode) on every chunk the real c-blosc 1.21 wrote: LZ4 /
    LZ4HC chunks with byte shuffle or none (and stored chunks) are planned, everything else is handed back as unsupported.
    The plan is then EXECUTED on the host with liblz4 (pyarrow's lz4_raw) + a numpy unshuffle — what `k_lz4_streams` and
    `k_unshuffle_blocks` do in HBM — and must reproduce the recipe bit for bit."""
    import pyarrow as pa
    chunk = base64.b64decode(case["chunk_b64"])
    raw = recipe(case["recipe"], case["n"], case["dtype"], case["seed"])
    info = codec.blosc_info(chunk)
    base, streams, blocks, tmpb, maxd, res = _plan_one(chunk, raw.nbytes)
    takes = info["stored"] or (case["cname"] in ("lz4", "lz4hc") and case["shuffle"] != 2)
    if not takes:
        assert res == codec.E_UNSUPPORTED and len(streams) == 0 and len(blocks) == 0
        return
    assert res == raw.nbytes
    out = np.zeros(raw.nbytes, dtype=np.uint8)
    tmp = np.zeros(max(tmpb, 1), dtype=np.uint8)
    covered = np.zeros(raw.nbytes, dtype=np.int32)
    for s in streams:
        src = base[s["src_off"]:s["src_off"] + s["csize"]].tobytes()
        dec = src if s["csize"] == s["dsize"] else pa.Codec("lz4_raw").decompress(src, decompressed_size=int(s["dsize"]), asbytes=True)
        assert len(dec) == s["dsize"]
        (out if s["to_out"] else tmp)[s["dst_off"]:s["dst_off"] + s["dsize"]] = np.frombuffer(dec, dtype=np.uint8)
        if s["to_out"]:
            covered[s["dst_off"]:s["dst_off"] + s["dsize"]] += 1
    for b in blocks:
        ts, bs = int(b["typesize"]), int(b["bsize"])
        n = bs // ts
        planes = tmp[b["tmp_off"]:b["tmp_off"] + n * ts].reshape(ts, n)
        out[b["out_off"]:b["out_off"] + n * ts] = planes.T.reshape(-1)
        out[b["out_off"] + n * ts:b["out_off"] + bs] = tmp[b["tmp_off"] + n * ts:b["tmp_off"] + bs]
        covered[b["out_off"]:b["out_off"] + bs] += 1
    assert (covered == 1).all()                                   # every output byte has exactly one producer
    assert out.tobytes() == raw.tobytes()


def test_gpu_decode_plan_refuses_damage_and_batches():
    lz4 = [c for c in CASES if c["cname"] in ("lz4", "lz4hc") and c["shuffle"] == 1 and c["nbytes"] >= 100000][:2]
    assert len(lz4) == 2
    chunks = [base64.b64decode(c["chunk_b64"]) for c in lz4] + [base64.b64decode(next(c for c in CASES if c["cname"] == "zstd")["chunk_b64"])]
    offs = np.concatenate([[0], np.cumsum([(len(c) + 63) // 64 * 64 for c in chunks])])
    base = np.zeros(offs[-1], dtype=np.uint8)
    for o, c in zip(offs, chunks):
        base[o:o + len(c)] = np.frombuffer(c, dtype=np.uint8)
    streams, blocks = np.zeros(4096, dtype=codec.LZ4_STREAM), np.zeros(1024, dtype=codec.SHUFFLE_BLOCK)
    nb = [lz4[0]["nbytes"], lz4[1]["nbytes"], 10 ** 6]
    ns, nbk, tmpb, maxd, res = codec.blosc_lz4_plan(base, offs[:3], [len(c) for c in chunks], [0, nb[0], nb[0] + nb[1]], nb, streams, blocks)
    assert list(res[:2]) == nb[:2] and res[2] == codec.E_UNSUPPORTED            # the zstd chunk goes back to the host
    assert ns > 0 and (streams["src_off"][:ns] < offs[2]).all()                 # ... and left no records behind
    assert (streams["dst_off"][:ns][streams["to_out"][:ns] == 0] < tmpb).all() and tmpb >= nb[0] + nb[1]
    bad = base.copy(); bad[offs[1] + 16:offs[1] + 20] = np.frombuffer((2 ** 31 - 1).to_bytes(4, "little"), dtype=np.uint8)
    with pytest.raises(codec.CodecError, match="malformed"):
        codec.blosc_lz4_plan(bad, offs[:2], [len(c) for c in chunks[:2]], [0, nb[0]], nb[:2], streams, blocks)
    with pytest.raises(codec.CodecError):
        codec.blosc_lz4_plan(base, offs[:1], [len(chunks[0])], [0], [100], streams, blocks)          # destination too small
    with pytest.raises(codec.CodecError):
        codec.blosc_lz4_plan(base, offs[:1], [len(chunks[0])], [0], nb[:1], streams[:2], blocks)     # record list too small


def test_read_packed_packs_ranges_of_files_back_to_back(tmp_path):
    """`afcodec_read_packed` (the decode-in-HBM route's reader): whole files and byte ranges land back to back at 64-byte steps,
    missing files and absent chunks take no room and report -100, sizes come from the files; a buffer that is too small and a
    range beyond its file are errors."""
    rng = np.random.default_rng(3)
    blobs = [rng.integers(0, 256, n, dtype=np.uint8) for n in (1, 63, 64, 3_000_001, 0, 200_000)]
    paths = []
    for i, b in enumerate(blobs):
        paths.append(str(tmp_path / f"c{i}"))
        b.tofile(paths[-1])
    locs = [(paths[0], 0, -1), (str(tmp_path / "absent"), 0, -1), (paths[1], 0, -1), None, (paths[2], 0, -1), (paths[3], 0, -1),
            (paths[3], 1_000_000, 1_500_000), (paths[4], 0, -1), (paths[5], 7, 199_000)]
    dst = np.full(6_000_000, 0xEE, dtype=np.uint8)
    offs, sizes = codec.read_packed(locs, dst, 64, threads=5)
    want = [blobs[0], None, blobs[1], None, blobs[2], blobs[3], blobs[3][1_000_000:2_500_000], blobs[4], blobs[5][7:199_007]]
    assert sizes.tolist() == [-100 if w is None else len(w) for w in want]
    assert offs[0] == 0 and (offs % 64 == 0).all() and len(offs) == len(locs) + 1
    for i, w in enumerate(want):
        room = 0 if w is None else (len(w) + 63) // 64 * 64
        assert offs[i + 1] - offs[i] == room
        if w is not None:
            assert np.array_equal(dst[offs[i]:offs[i] + len(w)], w)
    assert (dst[offs[-1]:] == 0xEE).all()
    with pytest.raises(codec.CodecError, match="do not fit"):
        codec.read_packed(locs, dst[:3_000_000], 64, threads=2)
    with pytest.raises(codec.CodecError):
        codec.read_packed([(paths[0], 0, 2)], dst, 64)
    o, z = codec.read_packed([], dst, 64)
    assert o.tolist() == [0] and len(z) == 0


def test_batches_larger_than_the_descriptor_limit_read_every_file(tmp_path):
    """A batch may hold thousands of chunk files (io.py: up to 4096 per request) while RLIMIT_NOFILE is 1024 on most hosts.
    The batch readers used to open every file of a batch first; past the limit `open` failed with EMFILE, the file was
    reported as ABSENT (-100) and the caller filled existing chunks with the fill value — no error, wrong weighted means.
    Now no descriptor is kept (one per thread and 1 MiB piece), and only "no such file" ever means absent."""
    import resource
    n = 400
    paths = []
    for i in range(n):
        p = tmp_path / f"c{i}"
        p.write_bytes(bytes([i % 251]) * 1024)
        paths.append(str(p))
    locs = [(p, 0, -1) for p in paths] + [(str(tmp_path / "never_written"), 0, -1), None]
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    resource.setrlimit(resource.RLIMIT_NOFILE, (64, hard))
    try:
        dst = np.zeros((n + 2) * 1024, dtype=np.uint8)
        off, res = codec.read_packed(locs, dst, align=64, threads=8)
        outs = [np.zeros(1024, dtype=np.uint8) for _ in locs]
        got = codec.decode_ranges("raw", locs, outs, threads=8, exact=False)
    finally:
        resource.setrlimit(resource.RLIMIT_NOFILE, (soft, hard))
    assert list(res[:n]) == [1024] * n and list(res[n:]) == [-100, -100]
    assert got[:n] == [1024] * n and got[n:] == [-100, -100]
    for i in range(n):
        assert dst[off[i]] == i % 251 and dst[off[i] + 1023] == i % 251 and outs[i][0] == i % 251 and outs[i][-1] == i % 251


def test_unreadable_file_is_an_error_not_an_absent_chunk(tmp_path):
    """EACCES / EISDIR and friends must raise; only ENOENT / ENOTDIR mean "this chunk was never written"."""
    d = tmp_path / "a_directory"
    d.mkdir()
    for fn in (lambda: codec.read_packed([(str(d), 0, -1)], np.zeros(64, dtype=np.uint8)),
               lambda: codec.decode_ranges("zlib", [(str(d), 0, -1)], [np.zeros(64, dtype=np.uint8)], exact=False)):
        with pytest.raises(codec.CodecError):
            fn()
    # a path THROUGH a regular file (ENOTDIR) is an absent chunk, like a missing file
    f = tmp_path / "plain"
    f.write_bytes(b"x" * 10)
    _, res = codec.read_packed([(str(f / "0.0.0"), 0, -1)], np.zeros(64, dtype=np.uint8))
    assert list(res) == [-100]


@pytest.mark.parametrize("n, typesize", [(1001, 4), (4099, 4), (70001, 8), (131, 8), (65536 * 4 + 3, 4), (1000003, 255)])
@pytest.mark.parametrize("blocksize", [0, 1001, 4096])
def test_encoder_keeps_the_bytes_beyond_the_last_whole_element(n, typesize, blocksize):
    """A buffer that is not a whole number of elements: the clamped block size is rounded down to whole elements again
    (c-blosc's compute_blocksize), the remainder travels in the short, unsplit last block — a split block used to drop it."""
    rng = np.random.default_rng(n)
    a = rng.integers(0, 255, n, dtype=np.uint8)
    a[: n // 2] = 7
    enc = codec.blosc_encode(a.tobytes(), typesize=typesize, shuffle=True, blocksize=blocksize)
    assert len(enc) <= codec.load().afcodec_blosc_bound(n, blocksize)
    assert bytes(codec.blosc_decode(enc)) == a.tobytes()
