#!/usr/bin/env python3
"""Host emulation, lane for lane, of `k_lz4_streams_vec` (aggfly_amd/csrc/afhip_lz4_kernels.h): the speculative parse of a
64-byte window in every lane, the token walk, the prefix sums, the rounds of 64 consecutive output bytes with the owner
lookup (scatter + prefix maximum), the near ring and the pending loop.  The generic path (one sequence at a time) is plain
Python.  Runs the plan of `afcodec_blosc_lz4_plan` on the real c-blosc fixtures and on chunks of the in-tree encoder and
compares with the recipe, so the kernel's control flow is checked for termination, bounds and results before a launch."""
import base64, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_blosc_fixtures import recipe
from aggfly_amd import codec

NEAR = int(os.environ.get("NEAR", "4096"))
LANE = np.arange(64)


class Bad(Exception):
    pass


def lane_get(v, idx):
    return v[np.asarray(idx) & 63]


def decode_stream(comp, src_off, csize, dsize, stats):
    src = comp[src_off:src_off + csize].astype(np.int64)
    if csize == dsize:
        return comp[src_off:src_off + csize].copy()
    if csize <= 0 or dsize <= 0:
        raise Bad("sizes")
    dst = np.full(dsize, 0xCD, np.int64); written = np.zeros(dsize, bool)
    near = np.zeros(NEAR, np.int64); near_pos = np.full(NEAR, -1, np.int64)       # near_pos: which position a slot holds (checks)

    def ring_write(pos, v):
        near[pos & (NEAR - 1)] = v; near_pos[pos & (NEAR - 1)] = pos

    def put(pos, v):
        assert 0 <= pos < dsize and not written[pos]
        dst[pos] = v; written[pos] = True; ring_write(pos, v)

    def line(k):
        a = np.minimum(k * 64 + LANE, csize - 1)
        return src[a]
    p = op = 0; iters = 0
    while True:
        iters += 1
        if p >= csize or iters > csize:
            raise Bad("ran off the stream")
        lk = p >> 6
        q0, q1 = line(lk), line(lk + 1)
        idx = (p & 63) + LANE
        w = np.where(idx < 64, lane_get(q0, idx), lane_get(q1, idx))
        wl = min(csize - p, 64)
        Mn = w & 15; extL = (w >> 4) == 15; extM = Mn == 15
        b1 = lane_get(w, LANE + 1)
        L = np.where(extL, 15 + b1, w >> 4)
        ls = LANE + np.where(extL, 2, 1)
        opos = ls + L
        off = lane_get(w, opos) | (lane_get(w, opos + 1) << 8)
        e = lane_get(w, opos + 2)
        M = np.where(extM, 19 + e, Mn + 4)
        nx = opos + np.where(extM, 3, 2)
        fits = ~(extL & (b1 == 255)) & ~(extM & (e == 255)) & (nx <= wl) & (p + nx < csize)
        nextv = np.where(fits, nx, 65)
        # the t-th token of the window by binary lifting (lane t), then the token lanes marked through LDS
        def hop(table, x):
            return np.where(x < 64, lane_get(table, x), 65)
        j2 = hop(nextv, nextv); j4 = hop(j2, j2); j8 = hop(j4, j4); j16 = hop(j8, j8)
        pos = np.where(LANE & 1, nextv[0], 0)
        for bit, tab in ((2, j2), (4, j4), (8, j8), (16, j16)):
            pos = np.where(LANE & bit, hop(tab, pos), pos)
        nxt = hop(nextv, pos)
        valid = (LANE < 32) & (nxt <= 64)
        tmask = np.zeros(64, bool); tmask[pos[valid]] = True
        n = int(tmask.sum()); c = int(nxt[n - 1]) if n else 0
        ref_mask = np.zeros(64, bool); rc = 0; rp = 0       # the serial walk it replaces
        while rp < 64:
            t = int(nextv[rp])
            if t > 64:
                break
            ref_mask[rp] = True; rp = rc = t
        assert (ref_mask == tmask).all() and rc == c, (ref_mask.nonzero(), tmask.nonzero(), rc, c)
        if tmask.any():
            ln = np.where(tmask, L + M, 0)
            incl = np.cumsum(ln); drel = incl - ln; tot = int(incl[63])
            if tot > dsize - op or (tmask & ((off == 0) | (off > op + drel + L))).any():
                raise Bad("group validation")
            recA = ls | (L << 6) | (drel << 13)
            carry = 0
            stats["groups"] += 1; stats["seqs"] += int(tmask.sum())
            for r0 in range(0, tot, 64):
                stats["rounds"] += 1
                mark = np.zeros(64, np.int64)
                sel = tmask & (drel >= r0) & (drel < r0 + 64)
                mark[(drel - r0)[sel]] = (LANE + 1)[sel]
                z = np.maximum(np.maximum.accumulate(mark), carry); carry = int(z[63])
                ra = lane_get(recA, z - 1); off_t = lane_get(off, z - 1)
                L_t = (ra >> 6) & 127
                rel = r0 + LANE - (ra >> 13)
                posv = op + r0 + LANE
                act = r0 + LANE < tot; is_lit = rel < L_t
                val = lane_get(w, (ra & 63) + rel)
                for i in np.nonzero(act & is_lit)[0]:
                    assert 0 <= rel[i] and (ra[i] & 63) + rel[i] < wl
                    ring_write(int(posv[i]), int(val[i]))
                i_in = rel - L_t
                with np.errstate(divide="ignore", invalid="ignore"):
                    srcpos = posv - i_in - off_t + np.where(i_in < off_t, i_in, np.where(off_t > 0, i_in % np.maximum(off_t, 1), 0))
                todo = act & ~is_lit
                far = todo & (srcpos < op + r0 + 64 - NEAR)
                fv = np.zeros(64, np.int64)
                for i in np.nonzero(far)[0]:
                    assert written[srcpos[i]]
                    fv[i] = dst[srcpos[i]]
                pend = todo.copy()
                while pend.any():
                    stats["passes"] += 1
                    frontier = op + r0 + int(np.nonzero(pend)[0][0])
                    ready = pend & (srcpos < frontier)
                    if not ready.any():
                        raise Bad("no lane ready")
                    vals = {}
                    for i in np.nonzero(ready)[0]:
                        if far[i]:
                            vals[i] = fv[i]
                        else:
                            sp = int(srcpos[i])
                            assert near_pos[sp & (NEAR - 1)] == sp, ("ring does not hold", sp, near_pos[sp & (NEAR - 1)], op, r0, i)
                            vals[i] = near[sp & (NEAR - 1)]
                    for i, v in vals.items():
                        val[i] = v; ring_write(int(posv[i]), int(v))
                    pend &= ~ready
                for i in np.nonzero(act)[0]:
                    assert not written[posv[i]]
                    dst[posv[i]] = val[i]; written[posv[i]] = True
            p += c; op += tot
            continue
        # ---- generic path, one sequence (scalar restatement) ----
        stats["generic"] += 1
        token = int(w[0]); Lg = token >> 4; q = p + 1
        if Lg == 15:
            while True:
                if q >= csize:
                    raise Bad("L ext")
                b = int(src[q]); q += 1; Lg += b
                if b != 255:
                    break
        if Lg > dsize - op or q + Lg > csize:
            raise Bad("literals")
        for i in range(Lg):
            put(op + i, int(src[q + i]))
        p = q + Lg; op += Lg
        if p >= csize:
            break
        if p + 2 > csize:
            raise Bad("offset")
        offg = int(src[p]) | (int(src[p + 1]) << 8); p += 2
        Mg = (token & 15) + 4
        if (token & 15) == 15:
            while True:
                if p >= csize:
                    raise Bad("M ext")
                b = int(src[p]); p += 1; Mg += b
                if b != 255:
                    break
        if offg == 0 or offg > op or Mg > dsize - op:
            raise Bad("match")
        for i in range(Mg):
            put(op + i, int(dst[op + i - offg]))
        op += Mg
    if op != dsize or not written.all():
        raise Bad("short output")
    return dst.astype(np.uint8)


def run_plan(chunks, nbytes, stats):
    offs = np.concatenate([[0], np.cumsum([(len(c) + 63) // 64 * 64 for c in chunks])]).astype(np.int64)
    base = np.zeros(max(int(offs[-1]), 64), np.uint8)
    for o, c in zip(offs, chunks):
        base[o:o + len(c)] = np.frombuffer(c, np.uint8)
    out_off = np.concatenate([[0], np.cumsum([(n + 63) // 64 * 64 for n in nbytes])]).astype(np.int64)
    streams, blocks = np.zeros(1 << 16, codec.LZ4_STREAM), np.zeros(1 << 14, codec.SHUFFLE_BLOCK)
    ns, nb, tmpb, maxd, res = codec.blosc_lz4_plan(base, offs[:-1], [len(c) for c in chunks], out_off[:-1], nbytes, streams, blocks)
    out = np.zeros(max(int(out_off[-1]), 64), np.uint8); tmp = np.zeros(max(tmpb, 64), np.uint8)
    for s in streams[:ns]:
        d = decode_stream(base, int(s["src_off"]), int(s["csize"]), int(s["dsize"]), stats)
        (out if s["to_out"] else tmp)[int(s["dst_off"]):int(s["dst_off"]) + len(d)] = d
    for b in blocks[:nb]:
        ts, bs = int(b["typesize"]), int(b["bsize"]); n = bs // ts
        t = tmp[int(b["tmp_off"]):int(b["tmp_off"]) + bs]
        o = out[int(b["out_off"]):int(b["out_off"]) + bs]
        o[:n * ts] = t[:n * ts].reshape(ts, n).T.reshape(-1); o[n * ts:] = t[n * ts:]
    return [out[o:o + n] if r >= 0 else None for o, n, r in zip(out_off[:-1], nbytes, res)]


if __name__ == "__main__":
    stats = dict(groups=0, seqs=0, rounds=0, passes=0, generic=0)
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "blosc_fixtures.json")))["cases"]
    chunks = [base64.b64decode(c["chunk_b64"]) for c in cases]
    raws = [recipe(c["recipe"], c["n"], c["dtype"], c["seed"]) for c in cases]
    got = run_plan(chunks, [r.nbytes for r in raws], stats)
    n_ok = 0
    for c, g, r in zip(cases, got, raws):
        if g is not None:
            assert g.tobytes() == r.tobytes(), (c["cname"], c["shuffle"], c["dtype"], c["recipe"], c["n"])
            n_ok += 1
    print("c-blosc fixtures decoded by the emulation:", n_ok, stats)
    rng = np.random.default_rng(5)
    for kind in ("smooth", "noisy", "constant", "random", "runs"):
        n = 150_000
        x = {"smooth": (280 + 10 * np.sin(np.arange(n) / 50)).astype("<f4"),
             "noisy": (280 + 10 * np.sin(np.arange(n) / 50) + rng.normal(0, 0.3, n)).astype("<f4"),
             "constant": np.full(n, 3.25, "<f4"), "random": rng.integers(0, 2 ** 31, n).astype("<i4").view("<f4"),
             "runs": np.repeat(rng.integers(0, 50, n // 100 + 1), 100)[:n].astype("<f4")}[kind]
        stats = dict(groups=0, seqs=0, rounds=0, passes=0, generic=0)
        for shuffle in (True, False):
            got = run_plan([codec.blosc_encode(x, 4, shuffle, 0)], [x.nbytes], stats)
            assert got[0].tobytes() == x.tobytes(), (kind, shuffle)
        print(kind, "ok", stats)
