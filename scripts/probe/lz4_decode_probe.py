#!/usr/bin/env python3
"""Where k_lz4_streams spends its time on one batch of bench-field chunks (56 chunks of 24 x 104 x 236 float32, Blosc-LZ4 +
shuffle, blocks split into byte planes): all streams, then only the stored planes, only the LZ4 planes (<= 64 KiB), only the
unsplit last blocks (259 KB each).  HIP-event times, best of 5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aggfly_amd import codec, hip, synth

field = sys.argv[1] if len(sys.argv) > 1 else "noisy"
T, ny, nx = 24 * 56, 104, 236
if field == "noisy":
    cube = synth.temperature_cube(T, ny, nx, dtype=np.float32, seed=1) + np.float32(273.15)
else:
    k = np.arange(T)[:, None, None]; y = np.arange(ny)[None, :, None]; x = np.arange(nx)[None, None, :]
    cube = (np.round((285 + 5 * np.sin(2 * np.pi * (k % 24) / 24) + 8 * np.sin(y / 17.0) * np.cos(x / 23.0)) * 100) / 100).astype(np.float32)
chunks = [codec.blosc_encode(cube[24 * i:24 * (i + 1)], 4, True, 0) for i in range(56)]
cb = cube[:24].nbytes
offs = np.concatenate([[0], np.cumsum([(len(c) + 63) // 64 * 64 for c in chunks])]).astype(np.int64)
base = np.zeros(int(offs[-1]), dtype=np.uint8)
for o, c in zip(offs, chunks):
    base[o:o + len(c)] = np.frombuffer(c, dtype=np.uint8)
streams, blocks = np.zeros(1 << 16, dtype=codec.LZ4_STREAM), np.zeros(1 << 14, dtype=codec.SHUFFLE_BLOCK)
ns, nb, tmpb, maxd, res = codec.blosc_lz4_plan(base, offs[:-1], [len(c) for c in chunks], np.arange(56) * cb, np.full(56, cb), streams, blocks)
st = streams[:ns]
print(f"field {field}: ratio {56 * cb / sum(len(c) for c in chunks):.2f}; {ns} streams: {(st['csize'] == st['dsize']).sum()} stored, "
      f"{((st['csize'] != st['dsize']) & (st['dsize'] <= 65536)).sum()} LZ4 planes, {(st['dsize'] > 65536).sum()} unsplit blocks; max dsize {maxd}")
comp = torch.from_numpy(base).cuda()
out = torch.zeros(56 * cb, dtype=torch.uint8, device="cuda"); tmp = torch.zeros(tmpb + 64, dtype=torch.uint8, device="cuda")
err = torch.zeros(1, dtype=torch.int32, device="cuda")


def run(sel, name):
    sub = np.ascontiguousarray(st[sel])
    if not len(sub):
        return
    dev = torch.from_numpy(sub.view(np.uint8).copy()).cuda()
    md = int(sub["dsize"].max())
    best = 1e9
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.lz4_decode_streams(comp, dev, len(sub), md, tmp, out, err); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    nbytes = int(sub["dsize"].sum())
    lz = sub["csize"] != sub["dsize"]
    print(f"{name:28s} {len(sub):5d} streams {nbytes / 1e6:8.1f} MB decoded  {best:7.3f} ms  {nbytes / best / 1e6:7.1f} GB/s"
          + (f"   (LZ4 input {int(sub['csize'][lz].sum()) / 1e6:.1f} MB)" if lz.any() else ""))


run(np.ones(ns, bool), "all")
run(st["csize"] == st["dsize"], "stored planes")
run((st["csize"] != st["dsize"]) & (st["dsize"] <= 65536), "LZ4 planes")
run(st["dsize"] > 65536, "unsplit last blocks")
big = np.nonzero(st["dsize"] > 65536)[0]
if len(big):
    run(np.isin(np.arange(ns), big[:1]), "ONE unsplit last block")
assert int(err.item()) == 0
