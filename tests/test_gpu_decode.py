"""Chunk decode in HBM (`afhip_lz4_decode_streams` + `afhip_unshuffle_blocks`, planned by `afcodec_blosc_lz4_plan`):
every LZ4 chunk the real c-blosc 1.21 wrote decodes bit-exact on the GPU; damaged streams are counted, never followed out of
bounds; stores read through `dataset_from_path(device="cuda")` give the same cube with the decode on the GPU or on the host."""
import base64
import json
import os
import sys

import numpy as np
import pandas as pd
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_blosc_fixtures import recipe          # noqa: E402

import aggfly_amd as af                         # noqa: E402
from aggfly_amd import codec, synth             # noqa: E402

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(HERE, "golden", "blosc_fixtures.json")))["cases"]


def _gpu_decode(torch, chunks, nbytes):
    """Plan + decode a batch of chunks on the GPU -> (list of decoded uint8 arrays | None for unsupported, error count)."""
    from aggfly_amd import hip
    offs = np.concatenate([[0], np.cumsum([(len(c) + 63) // 64 * 64 for c in chunks])]).astype(np.int64)
    base = np.zeros(max(int(offs[-1]), 64), dtype=np.uint8)
    for o, c in zip(offs, chunks):
        base[o:o + len(c)] = np.frombuffer(c, dtype=np.uint8)
    out_off = np.concatenate([[0], np.cumsum([(n + 63) // 64 * 64 for n in nbytes])]).astype(np.int64)
    streams, blocks = np.zeros(1 << 16, dtype=codec.LZ4_STREAM), np.zeros(1 << 14, dtype=codec.SHUFFLE_BLOCK)
    ns, nb, tmpb, maxd, res = codec.blosc_lz4_plan(base, offs[:-1], [len(c) for c in chunks], out_off[:-1], nbytes, streams, blocks)
    comp = torch.from_numpy(base).cuda()
    st = torch.from_numpy(streams[:max(ns, 1)].view(np.uint8).copy()).cuda()
    bl = torch.from_numpy(blocks[:max(nb, 1)].view(np.uint8).copy()).cuda()
    out = torch.full((max(int(out_off[-1]), 64),), 0xAB, dtype=torch.uint8, device="cuda")
    tmp = torch.zeros(max(tmpb, 64), dtype=torch.uint8, device="cuda")
    errors = torch.zeros(1, dtype=torch.int32, device="cuda")
    if ns:
        hip.lz4_decode_streams(comp, st, ns, maxd, tmp, out, errors)
    if nb:
        hip.unshuffle_blocks(tmp, out, bl, nb, int(blocks["bsize"][:nb].max()))
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    got = [host[o:o + n] if r >= 0 else None for o, n, r in zip(out_off[:-1], nbytes, res)]
    return got, int(errors.item()), host, out_off


def test_real_cblosc_lz4_chunks_decode_bit_exact_in_hbm(torch_cuda):
    chunks = [base64.b64decode(c["chunk_b64"]) for c in CASES]
    raws = [recipe(c["recipe"], c["n"], c["dtype"], c["seed"]) for c in CASES]
    got, nerr, host, out_off = _gpu_decode(torch_cuda, chunks, [r.nbytes for r in raws])
    assert nerr == 0
    taken = 0
    for c, g, r in zip(CASES, got, raws):
        info = codec.blosc_info(base64.b64decode(c["chunk_b64"]))
        if info["stored"] or (c["cname"] in ("lz4", "lz4hc") and c["shuffle"] != 2):
            assert g is not None
            assert g.tobytes() == r.tobytes(), (c["cname"], c["shuffle"], c["dtype"], c["recipe"], c["n"])
            taken += 1
        else:
            assert g is None
    assert taken >= 25
    # the 64-byte gaps between the outputs were never written
    for o, n, nxt in zip(out_off[:-1], [r.nbytes for r in raws], out_off[1:]):
        assert (host[o + n:nxt] == 0xAB).all()


@pytest.mark.parametrize("dtype,n,shuffle,blocksize", [("<f4", 1_000_000, True, 0), ("<f8", 333_333, True, 65536), ("<f4", 17, True, 0),
                                                        ("<i2", 50_000, False, 4096), ("<f4", 700_001, True, 10_000), ("<f4", 262_144, True, 262_144)])
def test_in_tree_encoder_chunks_decode_in_hbm(torch_cuda, dtype, n, shuffle, blocksize):
    """Multi-block chunks (up to 61 blocks x 4 byte planes), a short last block, tiny chunks, unshuffled int16, and data of
    every compressibility: smooth (long matches), noisy mantissas (stored planes), constant (maximal matches), random."""
    rng = np.random.default_rng(n)
    for kind in ("smooth", "noisy", "constant", "random", "runs"):
        if kind == "smooth":
            x = (280 + 10 * np.sin(np.arange(n) / 50)).astype(dtype)
        elif kind == "noisy":
            x = (280 + 10 * np.sin(np.arange(n) / 50) + rng.normal(0, 0.3, n)).astype(dtype)
        elif kind == "constant":
            x = np.full(n, 273.15).astype(dtype)
        elif kind == "random":
            x = np.frombuffer(rng.bytes(n * np.dtype(dtype).itemsize), dtype=dtype).copy()
        else:                                                        # long literal runs between long matches
            x = np.where((np.arange(n) // 700) % 2 == 0, 1.5, rng.normal(0, 1, n)).astype(dtype)
        enc = codec.blosc_encode(x, x.dtype.itemsize, shuffle, blocksize)
        assert codec.blosc_decode(enc).tobytes() == x.tobytes()
        got, nerr, _, _ = _gpu_decode(torch_cuda, [enc], [x.nbytes])
        assert nerr == 0 and got[0] is not None and got[0].tobytes() == x.tobytes(), kind


def test_damaged_streams_are_counted_not_followed(torch_cuda):
    """Garbage in the payload: the kernel must stop at the first impossible offset / length, bump the error counter and leave
    everything outside the stream's own destination untouched (the 0xAB guard bytes around the outputs)."""
    rng = np.random.default_rng(5)
    x = (280 + 10 * np.sin(np.arange(400_000) / 50) + rng.normal(0, 0.05, 400_000)).astype("<f4")
    enc = bytearray(codec.blosc_encode(x, 4, True, 0))
    good = bytes(enc)
    total_err = 0
    for trial in range(6):
        bad = bytearray(good)
        lo = 16 + 4 * 7 + 64
        for _ in range(40):                                          # overwrite runs of payload bytes (block table kept)
            p = int(rng.integers(lo, len(bad) - 16))
            bad[p:p + 8] = rng.bytes(8)
        try:
            got, nerr, host, out_off = _gpu_decode(torch_cuda, [good, bytes(bad), good], [x.nbytes] * 3)
        except codec.CodecError:
            continue                                                 # a stream-length prefix was hit: refused by the planner
        total_err += nerr
        assert got[0].tobytes() == x.tobytes() and got[2].tobytes() == x.tobytes()        # the neighbours are intact
        for o, nxt in zip(out_off[:-1] + x.nbytes, out_off[1:]):
            assert (host[o:nxt] == 0xAB).all()
    assert total_err > 0


def _store(tmp_path, name, cube, chunks, time=None):
    T, ny, nx = cube.shape
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"],
                                 {"time": time if time is not None else pd.date_range("2001-01-01", periods=T, freq="h"),
                                  "latitude": 30 + 0.25 * np.arange(ny), "longitude": 250 + 0.25 * np.arange(nx)}), lon_is_360=True)
    path = str(tmp_path / name)
    af.dataset_to_zarr(ds, path, var="t2m", chunks=chunks)
    return path


@pytest.mark.parametrize("layout", ["time_contiguous", "space_tiled", "whole_series_tiles"])
def test_store_to_hbm_gpu_decode_equals_host_decode(torch_cuda, tmp_path, monkeypatch, layout):
    """`dataset_from_path(device="cuda")` on Blosc-LZ4 stores: with the chunks decoded in HBM (AGGFLY_HIP_GPU_DECODE=1; the
    default for requests of 96 MB or more — 256 MB where the chunks hold whole time steps) or on the host threads (=0) the cube is the same, bit for bit — whole store,
    time windows that start and end inside chunks; the GPU route never calls the host Blosc decoder."""
    T, ny, nx = 24 * 30, 40, 64
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=3, ocean_frac=0.1, scattered_nan=40) + np.float32(273.15)
    chunks = {"time_contiguous": {"time": 48, "latitude": ny, "longitude": nx}, "space_tiled": {"time": 100, "latitude": 16, "longitude": 24},
              "whole_series_tiles": {"time": T, "latitude": 8, "longitude": 16}}[layout]
    path = _store(tmp_path, "s.zarr", cube, chunks)
    from aggfly_amd import io as afio
    kinds = []
    real, real_packed = codec.decode_ranges, codec.read_packed
    monkeypatch.setattr(codec, "decode_ranges", lambda kind, locs, outs, threads=8, exact=True: kinds.append(kind) or real(kind, locs, outs, threads, exact))
    monkeypatch.setattr(codec, "read_packed", lambda locs, dst, align=64, threads=8: kinds.append("files as they are") or real_packed(locs, dst, align, threads))
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "1")                 # (these stores are below the size the route starts at)
    dev = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda")
    assert set(kinds) == {"files as they are"}, kinds                # no host decode
    np.testing.assert_array_equal(dev.cube().cpu().numpy(), cube)
    win = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda", time_sel=slice("2001-01-05", "2001-01-11"))
    np.testing.assert_array_equal(win.cube().cpu().numpy(), cube[24 * 4:24 * 11])
    # small batches, a window that starts and ends inside chunks: batches of whole chunks decode straight into the cube, the
    # two edge batches go through the staging buffer
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_BATCH_MB", "1")
    win = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda", time_sel=slice("2001-01-02 06:00", "2001-01-21 19:00"))
    np.testing.assert_array_equal(win.cube().cpu().numpy(), cube[30:500])
    # ... and with the host-decoded tail that large requests have (here: the last fifth of the window's chunks, whatever the size):
    # the tail's chunks are placed like the host route's, also when the window ends inside them
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MIN_MB", "0")
    kinds.clear()
    for sel, want in ((None, cube), (slice("2001-01-02 06:00", "2001-01-21 19:00"), cube[30:500]), (slice("2001-01-05", "2001-01-11"), cube[24 * 4:24 * 11])):
        win = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda", time_sel=sel)
        np.testing.assert_array_equal(win.cube().cpu().numpy(), want)
    assert "blosc" in kinds and "files as they are" in kinds, kinds
    monkeypatch.delenv("AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MIN_MB")
    monkeypatch.delenv("AGGFLY_HIP_GPU_DECODE_BATCH_MB")
    kinds.clear()
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "0")
    host = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda")
    assert "blosc" in set(kinds)
    np.testing.assert_array_equal(host.cube().cpu().numpy(), dev.cube().cpu().numpy())


def test_float64_stores_decode_in_hbm(torch_cuda, tmp_path, monkeypatch):
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "1")
    T, ny, nx = 24 * 10, 12, 20
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float64, seed=4, scattered_nan=10)
    path = _store(tmp_path, "d.zarr", cube, {"time": 24, "latitude": ny, "longitude": nx})
    np.testing.assert_array_equal(af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda").cube().cpu().numpy(), cube)


def test_large_request_decodes_in_hbm_by_default(torch_cuda, tmp_path, monkeypatch):
    """No environment switch: a 306 MB request on a space-tiled store (>= `io.GPU_DECODE_AUTO_BYTES`) reads its chunk files as they
    are and decodes them on the GPU — several batches in flight — and gives the source cube; so does the same cube in chunks of whole
    time steps (`io.GPU_DECODE_AUTO_BYTES_WHOLE_ROWS`); small windows stay on the host threads in either layout."""
    from aggfly_amd import io as afio
    monkeypatch.delenv("AGGFLY_HIP_GPU_DECODE", raising=False)
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_BATCH_MB", "64")
    T, ny, nx = 24 * 130, 104, 236
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=11) + np.float32(273.15)
    assert afio.GPU_DECODE_AUTO_BYTES <= cube.nbytes and afio.GPU_DECODE_AUTO_BYTES_WHOLE_ROWS <= cube.nbytes
    path = _store(tmp_path, "big.zarr", cube, {"time": 240, "latitude": 52, "longitude": 118})
    kinds = []
    real, real_packed = codec.decode_ranges, codec.read_packed
    monkeypatch.setattr(codec, "decode_ranges", lambda kind, locs, outs, threads=8, exact=True: kinds.append(kind) or real(kind, locs, outs, threads, exact))
    monkeypatch.setattr(codec, "read_packed", lambda locs, dst, align=64, threads=8: kinds.append("files as they are") or real_packed(locs, dst, align, threads))
    dev = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda")
    # (space-tiled chunks: no host-decoded tail — that is for chunks of whole time steps, see test_store_to_hbm_gpu_decode_equals_host_decode)
    assert set(kinds) == {"files as they are"} and len(kinds) >= 5, kinds
    got = dev.cube().cpu().numpy()
    np.testing.assert_array_equal(got, cube)
    small = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda", time_sel=slice("2001-01-03", "2001-01-09"))
    assert kinds[-1] == "blosc"                                      # a window of one row of chunks (24 MB): host threads
    np.testing.assert_array_equal(small.cube().cpu().numpy(), cube[48:216])
    # the same cube in chunks of whole time steps: the same route from the same size on (round 2 kept such stores on the host threads up to
    # 768 MB), with the last chunks — a fifth of the request here — decoded by the host threads while the compressed batches upload
    rows = _store(tmp_path, "rows.zarr", cube, {"time": 24, "latitude": ny, "longitude": nx})
    kinds.clear()
    got = af.dataset_from_path(rows, "t2m", lon_is_360=True, device="cuda")
    assert set(kinds[:-1]) == {"files as they are"} and kinds[-1] == "blosc", kinds
    np.testing.assert_array_equal(got.cube().cpu().numpy(), cube)
    kinds.clear()
    small = af.dataset_from_path(rows, "t2m", lon_is_360=True, device="cuda", time_sel=slice("2001-01-03", "2001-02-09"))
    assert set(kinds) == {"blosc"}, kinds                            # 90 MB: host threads
    np.testing.assert_array_equal(small.cube().cpu().numpy(), cube[48:40 * 24])


@pytest.mark.parametrize("shards", [None, {"time": 96, "latitude": 24, "longitude": 64}])
def test_format3_stores_and_shards_decode_in_hbm(torch_cuda, tmp_path, monkeypatch, shards):
    """Zarr format 3 (``zarr.json``, ``c/`` keys), with and without ``sharding_indexed`` shards: the inner chunks of a shard are
    byte ranges of the shard file — the packed reader takes them as such — and absent chunks stay the fill value."""
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "1")
    T, ny, nx = 24 * 12, 24, 64
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=9, scattered_nan=25) + np.float32(273.15)
    ds = af.Dataset(af.DataArray(cube, ["time", "latitude", "longitude"], {"time": pd.date_range("2001-01-01", periods=T, freq="h"),
                                                                            "latitude": 30 + 0.25 * np.arange(ny), "longitude": 250 + 0.25 * np.arange(nx)}), lon_is_360=True)
    path = str(tmp_path / "v3.zarr")
    af.dataset_to_zarr(ds, path, var="t2m", chunks={"time": 48, "latitude": 12, "longitude": 32}, shards=shards, zarr_format=3)
    kinds = []
    real_packed = codec.read_packed
    monkeypatch.setattr(codec, "read_packed", lambda locs, dst, align=64, threads=8: kinds.append(len(locs)) or real_packed(locs, dst, align, threads))
    got = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda")
    assert kinds and sum(kinds) == (T // 48) * 2 * 2
    np.testing.assert_array_equal(got.cube().cpu().numpy(), cube)
    win = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda", time_sel=slice("2001-01-03 05:00", "2001-01-09 11:00"))
    np.testing.assert_array_equal(win.cube().cpu().numpy(), cube[53:204])


def test_chunk_of_another_geometry_sends_the_request_to_the_host_route(torch_cuda, tmp_path, monkeypatch):
    """`_gpu_decodable` judges a store by its FIRST chunk and sizes the route's buffers from it.  A later chunk written with
    another block size (or another inner codec) does not fit them / is not taken by the GPU decoder: the contract of
    include/aggfly_codec.h for such a chunk is "decode it on the host".  The request must then come out right through the
    host route — after the batches already in flight on the copy / kernel streams have drained — instead of raising
    mid-loop (round 2) or, worse, writing past the staging buffers."""
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", "1")
    monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_BATCH_MB", "1")         # several batches: the odd chunk is met mid-request
    T, ny, nx = 24 * 30, 40, 64
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=5, scattered_nan=40) + np.float32(273.15)
    path = _store(tmp_path, "odd.zarr", cube, {"time": 48, "latitude": ny, "longitude": nx})
    odd = os.path.join(path, "t2m", "9.0.0")
    assert os.path.exists(odd)
    with open(odd, "wb") as f:                                        # same data, 4 KiB blocks: 30x the streams the buffers were sized for
        f.write(codec.blosc_encode(cube[9 * 48:10 * 48].tobytes(), typesize=4, shuffle=True, blocksize=4096))
    kinds = []
    real, real_packed = codec.decode_ranges, codec.read_packed
    monkeypatch.setattr(codec, "decode_ranges", lambda kind, locs, outs, threads=8, exact=True: kinds.append(kind) or real(kind, locs, outs, threads, exact))
    monkeypatch.setattr(codec, "read_packed", lambda locs, dst, align=64, threads=8: kinds.append("files as they are") or real_packed(locs, dst, align, threads))
    got = af.dataset_from_path(path, "t2m", lon_is_360=True, device="cuda")
    assert "files as they are" in kinds and "blosc" in kinds, kinds   # started on the GPU route, finished on the host route
    np.testing.assert_array_equal(got.cube().cpu().numpy(), cube)


def test_two_threads_read_stores_at_the_same_time(torch_cuda, tmp_path, monkeypatch):
    """A process with a worker thread per job (or per GPU) reads stores concurrently: the page-locked staging buffers are cached per
    thread (`io._pinned_stage`), so neither route's uploads of one read can carry the other's bytes."""
    import threading
    from aggfly_amd import io as afio
    T, ny, nx = 24 * 20, 40, 64
    cubes = [synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=s, scattered_nan=10) + np.float32(273.15) for s in (21, 22)]
    paths = [_store(tmp_path, f"s{i}.zarr", c, {"time": 24, "latitude": ny, "longitude": nx}) for i, c in enumerate(cubes)]
    for mode in ("1", "0"):
        monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE", mode)
        monkeypatch.setenv("AGGFLY_HIP_GPU_DECODE_HOST_TAIL_MIN_MB", "0")
        bad, stages = [], {}

        def work(i):
            try:
                for _ in range(6):
                    got = af.dataset_from_path(paths[i], "t2m", lon_is_360=True, device="cuda").cube().cpu().numpy()
                    if not np.array_equal(got, cubes[i], equal_nan=True):
                        bad.append(i)
                stages[i] = {id(b) for (tid, _), bufs in afio._PINNED_STAGE.items() if tid == threading.get_ident() for b in bufs}
            except Exception as e:      # noqa: BLE001
                bad.append(repr(e))

        th = [threading.Thread(target=work, args=(i,)) for i in (0, 1)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not bad, bad
        assert stages[0] and stages[1] and not (stages[0] & stages[1])
