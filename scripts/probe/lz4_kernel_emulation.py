#!/usr/bin/env python3
"""Host emulation, lane for lane, of `k_lz4_streams` (aggfly_amd/csrc/afhip_lz4_kernels.h): the output ring with segment
flushes, the input ring with 1 KiB refills, the 64-byte parse window, the piecewise literal / match copies.  Runs the plan of
`afcodec_blosc_lz4_plan` on the real c-blosc fixtures and on chunks of the in-tree encoder (streams far longer than the ring)
and compares with the recipe: the kernel's control flow is checked for termination and bounds before it ever runs on a GPU."""
import base64, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_blosc_fixtures import recipe
from aggfly_amd import codec

SEG, RING_MAX, IN, PIECE = 8192, 65536 + 8192, 4096, 1024
LANES = np.arange(64)


class Bad(Exception):
    pass


def decode_stream(comp: np.ndarray, src_off: int, csize: int, dsize: int, ring: int, phase: int):
    """-> decoded bytes; raises Bad on a malformed stream.  ``phase`` stands in for the source address mod 16."""
    assert ring % SEG == 0 and SEG <= ring <= RING_MAX
    src = comp[src_off:src_off + csize]
    if csize == dsize:
        return src.copy()
    obuf = np.zeros(ring, np.uint8); ibuf = np.zeros(IN, np.uint8); dst = np.full(dsize, 0xCD, np.uint8)
    st = dict(units=0, p=0, op=0, opr=0, steps=0)

    def loaded():
        n = st["units"] * 16 - phase
        return min(max(n, 0), csize)

    def refill(upto):
        upto = min(upto, csize)
        while loaded() < upto:
            if (st["units"] + PIECE // 16) * 16 - IN > ((st["p"] + phase) & ~15):
                break
            for lane in range(64):
                u = st["units"] + lane
                if u * 16 < phase + csize:
                    lo = u * 16 - phase                      # stream offset of the unit's first byte (may be < 0 or run past the end)
                    for b in range(16):
                        q = lo + b
                        ibuf[(u * 16 + b) & (IN - 1)] = src[q] if 0 <= q < csize else 0xEE
            st["units"] += PIECE // 16

    def in_byte(q):
        q = np.asarray(q)
        assert ((q >= 0) & (q < loaded())).all() and (q >= st["units"] * 16 - phase - IN).all(), "input ring read outside the loaded window"
        return ibuf[(q + phase) & (IN - 1)].astype(np.int64)

    def flush(frm, n):
        r0 = frm % ring
        assert r0 + n <= ring and frm + n <= dsize
        dst[frm:frm + n] = obuf[r0:r0 + n]

    def advance(n):
        st["op"] += n; st["opr"] += n
        if st["opr"] >= ring:
            st["opr"] -= ring
        if st["op"] & (SEG - 1) == 0:
            flush(st["op"] - SEG, SEG)

    def room():
        return SEG - (st["op"] & (SEG - 1))

    def wrap(t):
        return np.where(t >= ring, t - ring, t)

    while True:
        st["steps"] += 1
        assert st["steps"] < 10 ** 7
        p = st["p"]
        if p >= csize:
            raise Bad("ran out of input")
        if loaded() < min(p + 64, csize):
            refill(p + 2048)
        at = p + LANES
        w = in_byte(np.where(at < csize, at, csize - 1))
        token = int(w[0]); L = token >> 4; hdr = 1
        if L == 15:
            q = p + 1
            while True:
                if q >= csize:
                    raise Bad("length extension past the end")
                if loaded() < min(q + 64, csize):
                    st["p"] = q; refill(q + 2048)
                a2 = q + LANES
                w2 = in_byte(np.where(a2 < csize, a2, csize - 1))
                not255 = (w2 != 255) | (a2 >= csize)
                k = int(np.argmax(not255)) if not255.any() else 64
                if k < 64:
                    if q + k >= csize:
                        raise Bad("length extension past the end")
                    L += 255 * k + int(w2[k]); q += k + 1
                    break
                L += 255 * 64; q += 64
                if L > dsize:
                    raise Bad("literal run longer than the stream")
            st["p"] = p = q; hdr = 64
            if L > dsize - st["op"] or p + L > csize:
                raise Bad("literal run out of range")
            left = L
            while left > 0:
                if loaded() <= st["p"]:
                    refill(st["p"] + 2048)
                n = min(loaded() - st["p"], left, room())
                assert n >= 1
                i = np.arange(n)
                obuf[wrap(st["opr"] + i)] = in_byte(st["p"] + i)
                st["p"] += n; left -= n; advance(n)
            p = st["p"]
        else:
            if L > dsize - st["op"] or p + 1 + L > csize:
                raise Bad("literal run out of range")
            n1 = min(L, room())
            if n1:
                obuf[wrap(st["opr"] + np.arange(n1))] = w[1:1 + n1]
                advance(n1)
            if L > n1:
                obuf[wrap(st["opr"] + np.arange(L - n1))] = w[1 + n1:1 + L]
                advance(L - n1)
            st["p"] = p = p + 1 + L
        if p >= csize:
            break
        if p + 2 > csize:
            raise Bad("offset past the end")
        if hdr + L + 2 <= 64:
            off = int(w[hdr + L]) | (int(w[hdr + L + 1]) << 8)
        else:
            if loaded() < p + 2:
                refill(p + 2048)
            o2 = in_byte(np.array([p, p + 1]))
            off = int(o2[0]) | (int(o2[1]) << 8)
        st["p"] = p = p + 2
        M = (token & 15) + 4
        if (token & 15) == 15:
            while True:
                if p >= csize:
                    raise Bad("length extension past the end")
                if loaded() < min(p + 64, csize):
                    st["p"] = p; refill(p + 2048)
                a2 = p + LANES
                w2 = in_byte(np.where(a2 < csize, a2, csize - 1))
                not255 = (w2 != 255) | (a2 >= csize)
                k = int(np.argmax(not255)) if not255.any() else 64
                if k < 64:
                    if p + k >= csize:
                        raise Bad("length extension past the end")
                    M += 255 * k + int(w2[k]); p += k + 1
                    break
                M += 255 * 64; p += 64
                if M > dsize:
                    raise Bad("match longer than the stream")
            st["p"] = p
        op = st["op"]
        if off == 0 or off > op or M > dsize - op:
            raise Bad("match out of range")
        left = M
        while left > 0:
            n = min(left, room())
            sr = st["opr"] - off
            if sr < 0:
                sr += ring
            for s0 in range(0, n, 64):                       # one wave instruction per 64 bytes, in order
                i = np.arange(s0, min(s0 + 64, n))
                a = wrap(sr + (i if off >= 64 else i % off))
                obuf[wrap(st["opr"] + i)] = obuf[a]
            left -= n; advance(n)
    if st["op"] != dsize:
        raise Bad(f"decoded {st['op']} of {dsize} bytes")
    if st["op"] & (SEG - 1):
        flush(st["op"] & ~(SEG - 1), st["op"] & (SEG - 1))
    return dst


def run_plan(chunk: bytes, nbytes: int, phase0=0):
    base = np.frombuffer(chunk, dtype=np.uint8).copy()
    streams, blocks = np.zeros(1 << 14, dtype=codec.LZ4_STREAM), np.zeros(1 << 12, dtype=codec.SHUFFLE_BLOCK)
    ns, nb, tmpb, maxd, res = codec.blosc_lz4_plan(base, [0], [len(chunk)], [0], [nbytes], streams, blocks)
    if res[0] < 0:
        return None
    ring = min(RING_MAX, max(SEG, (maxd + SEG - 1) // SEG * SEG))
    out, tmp = np.zeros(nbytes, np.uint8), np.zeros(max(tmpb, 1), np.uint8)
    for s in streams[:ns]:
        dec = decode_stream(base, int(s["src_off"]), int(s["csize"]), int(s["dsize"]), ring, (int(s["src_off"]) + phase0) & 15)
        (out if s["to_out"] else tmp)[s["dst_off"]:s["dst_off"] + s["dsize"]] = dec
    for b in blocks[:nb]:
        ts, bs = int(b["typesize"]), int(b["bsize"]); n = bs // ts
        out[b["out_off"]:b["out_off"] + n * ts] = tmp[b["tmp_off"]:b["tmp_off"] + n * ts].reshape(ts, n).T.reshape(-1)
        out[b["out_off"] + n * ts:b["out_off"] + bs] = tmp[b["tmp_off"] + n * ts:b["tmp_off"] + bs]
    return out


if __name__ == "__main__":
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "blosc_fixtures.json")))["cases"]
    n_ok = 0
    for c in cases:
        chunk = base64.b64decode(c["chunk_b64"])
        raw = recipe(c["recipe"], c["n"], c["dtype"], c["seed"])
        got = run_plan(chunk, raw.nbytes, phase0=len(chunk) % 16)
        if got is None:
            continue
        assert got.tobytes() == raw.tobytes(), (c["cname"], c["shuffle"], c["dtype"], c["recipe"], c["n"])
        n_ok += 1
    print("fixtures decoded through the emulated kernel:", n_ok)
    rng = np.random.default_rng(1)
    for dtype, n, shuffle, bs in (("<f4", 300_000, True, 0), ("<f8", 100_000, True, 65536), ("<i2", 50_000, False, 4096), ("<f4", 200_001, True, 0), ("<f4", 262144 // 4 * 3, False, 0)):
        for kind in ("smooth", "noisy", "constant", "random", "runs"):
            x = {"smooth": (280 + 10 * np.sin(np.arange(n) / 50)), "noisy": 280 + 10 * np.sin(np.arange(n) / 50) + rng.normal(0, 0.3, n),
                 "constant": np.full(n, 273.15), "random": rng.normal(0, 1e30, n), "runs": np.where((np.arange(n) // 700) % 2 == 0, 1.5, rng.normal(0, 1, n))}[kind].astype(dtype)
            enc = codec.blosc_encode(x, x.dtype.itemsize, shuffle, bs)
            got = run_plan(enc, x.nbytes, phase0=int(rng.integers(0, 16)))
            assert got is not None and got.tobytes() == x.tobytes(), (dtype, n, kind)
    print("in-tree encoder chunks (streams up to 256 KiB through the 72 KiB ring): ok")
    # damage: must raise Bad or return wrong bytes, never assert / index out of range
    x = (280 + 10 * np.sin(np.arange(200_000) / 50) + rng.normal(0, 0.05, 200_000)).astype("<f4")
    good = codec.blosc_encode(x, 4, True, 0)
    outcomes = {"bad": 0, "wrong": 0, "plan": 0}
    for t in range(30):
        bad = bytearray(good)
        for _ in range(20):
            q = int(rng.integers(16 + 4 * 4 + 8, len(bad) - 16)); bad[q:q + 8] = rng.bytes(8)
        try:
            got = run_plan(bytes(bad), x.nbytes)
            outcomes["wrong" if got is None or got.tobytes() != x.tobytes() else "bad"] += 0 if got is not None and got.tobytes() == x.tobytes() else 1
        except Bad:
            outcomes["bad"] += 1
        except codec.CodecError:
            outcomes["plan"] += 1
    print("damaged chunks:", outcomes)
