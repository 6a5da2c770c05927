// afhip_lz4_kernels.h — chunk decode in HBM for the ingestion path ("or GPU-side decode").
//
// The reference decodes Zarr chunks on host threads inside its dask graph (aggfly/dataset/dataset.py:697-728; read + decode is
// ~80 % of its end-to-end time, benchmarks/bench_read_scheduler.py:4-8).  Here a Blosc-1 chunk whose streams are LZ4 crosses
// PCIe COMPRESSED; the host only parses the container (block table, stream lengths: afcodec_blosc_lz4_plan in
// aggfly_amd/csrc/blosc1.c) and the GPU does the rest:
//
//   k_lz4_streams_vec  one wave per LZ4 stream (a Blosc block, or one byte plane of a split block; any length), the output
//                      written straight to its destination; see the comment at the kernel.
//   k_unshuffle_blocks Blosc's byte shuffle undone per block: element i's byte j sits at plane j, position i.
//
// Two earlier kernels are in the history of this file (round 2): one that decoded inside a 64 KiB LDS ring (one or two waves
// per CU: 4-5 GB/s of LZ4 output for the chip) and a scalar, sequence-by-sequence LDS-free one (~2,000 mostly scalar
// instructions per window of 8 sequences, bound by the CU's single scalar unit: 10-26 GB/s); profiles/README.md lists
// their measurements.
//
// Malformed streams (offsets before the start, lengths beyond the recorded sizes) never write outside the stream's own
// destination: the wave stops and bumps the error counter, which the host reads at its next synchronisation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afhip {

struct Lz4Stream {          // == afhip_lz4_stream (include/aggfly_hip.h)
    int64_t src_off;        // compressed bytes: comp + src_off
    int64_t dst_off;        // decoded bytes: (to_out ? out : tmp) + dst_off
    int32_t csize, dsize;   // csize == dsize: stored, copied as is
    int32_t to_out, pad;
};

struct ShufBlock {          // == afhip_shuffle_block
    int64_t tmp_off, out_off;
    int32_t bsize, typesize;
};

// A wave's stores and its later loads of the same bytes are ordered by s_waitcnt vmcnt(0) (the store has reached L2), and the
// loads are L2-served (agent-scope relaxed atomics = sc1: the per-CU L1 is not kept coherent with the stores).
__device__ __forceinline__ int ld_l2_u8(const uint8_t* p) {
    // 32-bit aligned container load, L2-served; the byte is extracted in registers
    const uint32_t* q = (const uint32_t*)((uintptr_t)p & ~(uintptr_t)3);
    const uint32_t v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)((v >> (8 * ((uintptr_t)p & 3))) & 0xff);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz4_streams_vec.  LZ4 is a chain of sequences (token, literals, offset, match length), each placed after the one before:
// resolved one at a time, a wave spends ~250 mostly scalar instructions per sequence, and a CU has ONE scalar unit.  Here
// the per-sequence work sits in the lanes:
//   * the compressed bytes arrive a 64-byte line of the stream at a time, loaded three lines ahead (their addresses do
//     not depend on the parse); a window = the 64 bytes at the read position, cut out of two lines by bpermute;
//   * every lane parses the window AS IF a token started at its byte (token, one optional extension byte per length,
//     offset: four bpermutes), which gives `next`, the lane of the following token; the real tokens are 0, next[0],
//     next[next[0]], ...: lane t finds the t-th of them by binary lifting (token_walk_lifted);
//   * a DPP prefix sum over the token lanes places every sequence in the output;
//   * the output is then produced 64 CONSECUTIVE bytes a round, a byte per lane: the lane finds the sequence it belongs
//     to (starts scattered through 256 B of LDS, prefix maximum), takes its literal from the window by bpermute or its
//     match byte from the last 4 KiB of output, mirrored in LDS (older history: back from L2), and the round leaves as
//     one coalesced store.  A match that reads bytes produced in the same round waits for the lanes before it (a loop
//     that retires at least the first pending lane per pass).
// Sequences that do not fit a window (literal runs of 60+ bytes, length extensions of more than one byte, the stream's
// last sequence) take the generic path at the end of the loop, one at a time.
// Measured (profiles/r02_lz4_vec_phase_cycles.txt, s_memtime per phase): ~6,400 cycles per window of ~18 sequences on the
// noisy byte planes of the bench field — parse 14 %, token walk 18 % (29 % of 7,400 with the serial walk), scan 5 %, owner
// lookup + gathers 37 %, pending loop 14 %, store 3 %, generic 9 % — i.e. 1.9 ms for a 64 KiB plane, start to end, whatever
// else the chip does: a launch takes that long once it has fewer streams than wave slots, which is why the ingest path
// feeds it large batches.
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or0(int v) {                   // the DPP-selected lane's v, 0 where there is none / masked off
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_scan_add(int v) {             // inclusive prefix sum over the 64 lanes
    v += dpp_or0<0x111, 0xf>(v);                                  // row_shr:1, 2, 4, 8: Hillis-Steele inside each row of 16
    v += dpp_or0<0x112, 0xf>(v);
    v += dpp_or0<0x114, 0xf>(v);
    v += dpp_or0<0x118, 0xf>(v);
    v += dpp_or0<0x142, 0xa>(v);                                  // row_bcast:15 -> rows 1, 3
    v += dpp_or0<0x143, 0xc>(v);                                  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ int wave_scan_max(int v) {             // inclusive prefix maximum, v >= 0
    v = max(v, dpp_or0<0x111, 0xf>(v));
    v = max(v, dpp_or0<0x112, 0xf>(v));
    v = max(v, dpp_or0<0x114, 0xf>(v));
    v = max(v, dpp_or0<0x118, 0xf>(v));
    v = max(v, dpp_or0<0x142, 0xa>(v));
    v = max(v, dpp_or0<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ int lane_get(int v, int from_lane) {   // v of lane (from_lane mod 64)
    return __builtin_amdgcn_ds_bpermute(from_lane << 2, v);
}
// i mod m, 0 <= i < 4096, 1 <= m < 65536: the float quotient is within one of the true one
__device__ __forceinline__ int mod_small(int i, int m) {
    const int q = (int)((float)i * __builtin_amdgcn_rcpf((float)m));
    int r = i - q * m;
    r += r < 0 ? m : 0;
    r -= r >= m ? m : 0;
    return r;
}

// The tokens of a window: 0, next[0], next[next[0]], ... while the sequence fits (next <= 64; 65 = it does not).
// -> bit mask of the token lanes; c = the byte after the last of them.  Serial form: one v_readlane round trip per sequence
// (2,100 of the 7,400 cycles a window took; kept as the reference of scripts/probe/dpp_scan_check.hip).
__device__ __forceinline__ uint64_t token_walk_serial(int nextv, int& c) {
    uint64_t tmask = 0;
    c = 0;
    for (int pos = 0; pos < 64;) {
        const int t = __builtin_amdgcn_readlane(nextv, pos);
        if (t > 64) break;
        tmask |= 1ull << pos;
        pos = c = t;
    }
    return tmask;
}
// Lifted form: lane t finds the t-th token by binary lifting (next^2, ^4, ^8, ^16 by bpermute; a window holds at most 21
// sequences), then the token lanes are marked through LDS (mark: 64 ints).  Ten bpermutes, six of them in a dependent
// chain.  Every lane takes every hop and keeps it or not: a bpermute inside a divergent branch reads 0 from the lanes the
// branch switched off.
__device__ __forceinline__ uint64_t token_walk_lifted(int nextv, int lane, volatile int* mark, int& c) {
    auto hop = [&](int table, int x) { const int y = lane_get(table, x); return x < 64 ? y : 65; };
    const int j2 = hop(nextv, nextv), j4 = hop(j2, j2), j8 = hop(j4, j4), j16 = hop(j8, j8);
    const int next0 = __builtin_amdgcn_readlane(nextv, 0);       // (taken before the select: inside it, "first lane" is lane 1)
    int pos = (lane & 1) ? next0 : 0;
    const int h2 = hop(j2, pos);
    pos = (lane & 2) ? h2 : pos;
    const int h4 = hop(j4, pos);
    pos = (lane & 4) ? h4 : pos;
    const int h8 = hop(j8, pos);
    pos = (lane & 8) ? h8 : pos;
    const int h16 = hop(j16, pos);
    pos = (lane & 16) ? h16 : pos;
    const int nxt = hop(nextv, pos);                              // where the t-th token's sequence ends (65: it does not fit)
    const bool valid = lane < 32 && nxt <= 64;
    mark[lane] = 0;
    if (valid) mark[pos] = 1;
    const uint64_t tmask = __builtin_amdgcn_ballot_w64(mark[lane] != 0);
    const int n = __builtin_popcountll(tmask);
    c = n ? __builtin_amdgcn_readlane(nxt, n - 1) : 0;
    return tmask;
}

constexpr int LZ4_NEAR = 4096;     // output bytes mirrored in LDS (2 / 8 / 16 KiB measured the same: profiles/r02_lz4_vec_near_ring_sizes.txt)

// PROF: cycle counts per phase and stream (s_memtime) into prof[stream * 8 + phase]; a measuring aid (AFHIP_LZ4_PROF=1)
template <int NEAR, bool PROF = false>
__global__ __launch_bounds__(64) void k_lz4_streams_vec(const uint8_t* __restrict__ comp, const Lz4Stream* __restrict__ streams,
                                                        uint8_t* tmp, uint8_t* out, int32_t* __restrict__ errors,
                                                        long long* __restrict__ prof = nullptr) {
    long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    auto tick = [&](int phase) {                                  // time since the last tick goes to `phase`
        if (PROF) {
            const long long t = __builtin_readcyclecounter();
            pc[phase] += t - t_last;
            t_last = t;
        }
    };
    if (PROF) t_last = __builtin_readcyclecounter();
    // Lanes talk to one another through these two arrays with no barrier in between (LDS operations of a wave execute in
    // order); volatile, because to the compiler a lane that stores 0 to mark[lane] and reads it back has read 0
    __shared__ volatile uint8_t near[NEAR];                      // output position q at near[q mod NEAR]
    __shared__ volatile int mark[64];                            // sequence starts of the round being produced
    const Lz4Stream s = streams[blockIdx.x];
    const int lane = threadIdx.x;
    const uint8_t* __restrict__ src = comp + s.src_off;
    uint8_t* dst = (s.to_out ? out : tmp) + s.dst_off;
    const int csize = s.csize, dsize = s.dsize;
    if (csize == dsize) {
        if (((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) {
            const int body = dsize & ~15;
            for (int i = lane * 16; i < body; i += 64 * 16) *(uint4*)(dst + i) = *(const uint4*)(src + i);
            for (int i = body + lane; i < dsize; i += 64) dst[i] = src[i];
        } else {
            for (int i = lane; i < dsize; i += 64) dst[i] = src[i];
        }
        return;
    }
    if (csize <= 0 || dsize <= 0) {
        if (lane == 0) atomicAdd(errors, 1);
        return;
    }
    auto put = [&](int pos, int v) {
        dst[pos] = (uint8_t)v;
        near[pos & (NEAR - 1)] = (uint8_t)v;
    };
    // the compressed bytes arrive a 64-byte line of the stream at a time, three lines ahead of the parse: the addresses do
    // not depend on what the parse finds
    auto ld_line = [&](int k) {
        const int a = k * 64 + lane;
        return (int)src[a < csize ? a : csize - 1];
    };
    int lk = 0, q0 = ld_line(0), q1 = ld_line(1), q2 = ld_line(2);
    int p = 0, op = 0, acked = 0, iters = 0;
    bool bad = false;
    while (true) {
        if (p >= csize || ++iters > csize) { bad = true; break; }
        if ((p >> 6) - lk > 2) {                                  // (a long literal run jumped ahead)
            lk = p >> 6;
            q0 = ld_line(lk);
            q1 = ld_line(lk + 1);
            q2 = ld_line(lk + 2);
        }
        while ((p >> 6) > lk) {
            q0 = q1;
            q1 = q2;
            ++lk;
            q2 = ld_line(lk + 2);
        }
        int w;                                                    // the window: stream byte p + lane
        {
            const int idx = (p & 63) + lane;
            const int a = lane_get(q0, idx), b = lane_get(q1, idx);
            w = idx < 64 ? a : b;
        }
        const int wl = (csize - p) < 64 ? (csize - p) : 64;      // valid bytes of the window
        // ---- every lane parses as if a token started at its byte ----
        const int Mn = w & 15;
        const bool extL = (w >> 4) == 15, extM = Mn == 15;
        const int b1 = lane_get(w, lane + 1);
        const int L = extL ? 15 + b1 : (w >> 4);
        const int ls = lane + (extL ? 2 : 1);                    // first literal (window index)
        const int opos = ls + L;                                 // the offset field
        const int off = lane_get(w, opos) | (lane_get(w, opos + 1) << 8);
        const int e = lane_get(w, opos + 2);
        const int M = extM ? 19 + e : Mn + 4;
        const int nx = opos + (extM ? 3 : 2);                    // the next token
        // whole inside the window, no second extension byte, and not the stream's end (its last sequence has no match)
        const bool fits = !(extL && b1 == 255) && !(extM && e == 255) && nx <= wl && p + nx < csize;
        const int nextv = fits ? nx : 65;
        if (PROF) { pc[6] += __builtin_amdgcn_readfirstlane(nextv) & 0; tick(0); }     // (the use makes the parse finish before the tick)
        int c = 0;
        const uint64_t tmask = token_walk_lifted(nextv, lane, mark, c);
        tick(1);
        if (tmask) {
            if (PROF) pc[7] += 1;
            const bool is_tok = (tmask >> lane) & 1;
            const int len = is_tok ? L + M : 0;
            const int incl = wave_scan_add(len);
            const int drel = incl - len;                          // the sequence's first output byte, relative to op
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (tot > dsize - op || __builtin_amdgcn_ballot_w64(is_tok && (off == 0 || off > op + drel + L))) { bad = true; break; }
            const int recA = ls | (L << 6) | (drel << 13);        // ls < 64, L < 64, drel < 2^13
            int carry = 0;
            tick(2);
            for (int r0 = 0; r0 < tot && !bad; r0 += 64) {
                // the sequence of every byte of the round: starts scattered to their lanes, then a prefix maximum
                mark[lane] = 0;
                if (is_tok && drel >= r0 && drel < r0 + 64) mark[drel - r0] = lane + 1;
                int z = wave_scan_max(mark[lane]);
                z = z > carry ? z : carry;
                carry = __builtin_amdgcn_readlane(z, 63);
                const int ra = lane_get(recA, z - 1), off_t = lane_get(off, z - 1);
                const int L_t = (ra >> 6) & 127;
                const int rel = r0 + lane - (ra >> 13);           // byte of the sequence: literals first, then the match
                const int pos = op + r0 + lane;
                const bool act = r0 + lane < tot, is_lit = rel < L_t;
                int val = lane_get(w, (ra & 63) + rel);          // (a literal lane's byte)
                if (act && is_lit) near[pos & (NEAR - 1)] = (uint8_t)val;
                const int i_in = rel - L_t;
                const int srcpos = pos - i_in - off_t + (i_in < off_t ? i_in : mod_small(i_in, off_t));
                const bool todo = act && !is_lit;
                // older history than the ring keeps (it is overwritten up to this round's end) comes back through L2
                const bool far = todo && srcpos < op + r0 + 64 - NEAR;
                if (__builtin_amdgcn_ballot_w64(far && srcpos >= acked)) {
                    __builtin_amdgcn_s_waitcnt(0);               // this wave's earlier stores have reached L2
                    acked = op + r0;
                }
                int fv = 0;
                if (far) fv = ld_l2_u8(dst + srcpos);
                uint64_t pend = __builtin_amdgcn_ballot_w64(todo);
                tick(3);
                while (pend) {
                    // everything before the first pending lane is in the ring (or on its way from L2 in fv)
                    const int frontier = op + r0 + __builtin_ctzll(pend);
                    const bool ready = ((pend >> lane) & 1) && srcpos < frontier;
                    if (ready) {
                        val = far ? fv : (int)near[srcpos & (NEAR - 1)];
                        near[pos & (NEAR - 1)] = (uint8_t)val;
                    }
                    const uint64_t done = __builtin_amdgcn_ballot_w64(ready);
                    if (!done) { bad = true; break; }             // (cannot happen: off >= 1 was checked)
                    pend &= ~done;
                }
                tick(4);
                if (act) dst[pos] = (uint8_t)val;
                tick(5);
            }
            if (bad) break;
            p += c;
            op += tot;
            continue;
        }
        // ---- generic path: one sequence with long length extensions, long copies, or the stream's last sequence ----
        const int token = __builtin_amdgcn_readlane(w, 0);
        int Lg = token >> 4, hdr = 1;
        bool window_ok = true;                                    // the offset still sits in the window
        if (Lg == 15) {
            int q = p + 1;
            while (true) {
                if (q >= csize) { bad = true; break; }
                const int a2 = q + lane;
                const int w2 = src[a2 < csize ? a2 : csize - 1];
                const uint64_t not255 = __builtin_amdgcn_ballot_w64(w2 != 255 || a2 >= csize);
                const int k = not255 ? __builtin_ctzll(not255) : 64;
                if (k < 64) {
                    if (q + k >= csize) { bad = true; break; }
                    Lg += 255 * k + __builtin_amdgcn_readlane(w2, k);
                    q += k + 1;
                    break;
                }
                Lg += 255 * 64;
                q += 64;
                if (Lg > dsize) { bad = true; break; }
            }
            if (bad) break;
            hdr = q - p;
            window_ok = false;
        }
        if (Lg > dsize - op || p + hdr + Lg > csize) { bad = true; break; }
        if (window_ok && 1 + Lg <= wl) {
            if (lane >= 1 && lane < 1 + Lg) put(op + lane - 1, w);
        } else {
            const uint8_t* from = src + p + hdr;
            for (int i = lane; i < Lg; i += 64) put(op + i, from[i]);
        }
        p += hdr + Lg;
        op += Lg;
        if (p >= csize) break;
        if (p + 2 > csize) { bad = true; break; }
        int offg;
        if (window_ok && 1 + Lg + 2 <= wl) {
            offg = __builtin_amdgcn_readlane(w, 1 + Lg) | (__builtin_amdgcn_readlane(w, 2 + Lg) << 8);
        } else {
            const int o2 = src[p + (lane & 1)];
            offg = __builtin_amdgcn_readlane(o2, 0) | (__builtin_amdgcn_readlane(o2, 1) << 8);
        }
        p += 2;
        int Mg = (token & 15) + 4;
        if ((token & 15) == 15) {
            while (true) {
                if (p >= csize) { bad = true; break; }
                const int a2 = p + lane;
                const int w2 = src[a2 < csize ? a2 : csize - 1];
                const uint64_t not255 = __builtin_amdgcn_ballot_w64(w2 != 255 || a2 >= csize);
                const int k = not255 ? __builtin_ctzll(not255) : 64;
                if (k < 64) {
                    if (p + k >= csize) { bad = true; break; }
                    Mg += 255 * k + __builtin_amdgcn_readlane(w2, k);
                    p += k + 1;
                    break;
                }
                Mg += 255 * 64;
                p += 64;
                if (Mg > dsize) { bad = true; break; }
            }
            if (bad) break;
        }
        if (offg == 0 || offg > op || Mg > dsize - op) { bad = true; break; }
        if (offg <= NEAR - 64) {
            // a near match of any length, 64 bytes a step through the ring.  off >= 64: a step reads bytes of earlier steps /
            // sequences only.  Shorter periods: the first step repeats the last `off` bytes; later steps copy from P bytes
            // back, P = the multiple of `off` in [64, 64 + off): written by earlier steps, never further back than the ring
            // keeps even for matches much longer than the ring
            const int P = offg >= 64 ? offg : offg * ((64 + offg - 1) / offg);
            for (int i0 = 0; i0 < Mg; i0 += 64) {
                const int i = i0 + lane;
                if (i < Mg) put(op + i, near[((i0 == 0 && offg < 64) ? op - offg + lane % offg : op + i - P) & (NEAR - 1)]);
            }
        } else {
            if (op - offg + (Mg < offg ? Mg : offg) > acked) {
                __builtin_amdgcn_s_waitcnt(0);
                acked = op;
            }
            const uint8_t* from = dst + op - offg;
            if (offg >= Mg) {                                     // no overlap: all loads first, then the stores
                for (int i = lane; i < Mg; i += 64) put(op + i, ld_l2_u8(from + i));
            } else {
                // overlapping by whole steps (off > NEAR - 64 >= 64): each step's source was stored by an earlier step
                for (int i0 = 0; i0 < Mg; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < Mg) put(op + i, ld_l2_u8(from + i));
                    __builtin_amdgcn_s_waitcnt(0);
                }
                acked = op + Mg;
            }
        }
        op += Mg;
        tick(6);
    }
    if (PROF && lane == 0)
        for (int i = 0; i < 8; ++i) prof[(size_t)blockIdx.x * 8 + i] = pc[i];
    if (bad || op != dsize) {
        if (lane == 0) atomicAdd(errors, 1);
    }
}

// Blosc's byte shuffle undone (unshuffle_bytes in blosc1.c): out[i * ts + j] = tmp[j * n + i], n = bsize / ts; the
// bsize % ts trailing bytes are copied as they are.  grid = (element tiles, blocks); a thread assembles one element.
__global__ __launch_bounds__(256) void k_unshuffle_blocks(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ out,
                                                          const ShufBlock* __restrict__ blocks) {
    const ShufBlock b = blocks[blockIdx.y];
    const int ts = b.typesize;
    const int64_t n = b.bsize / ts;
    const uint8_t* __restrict__ src = tmp + b.tmp_off;
    uint8_t* __restrict__ dst = out + b.out_off;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (ts == 4) {
            const uint32_t v = (uint32_t)src[i] | ((uint32_t)src[n + i] << 8) | ((uint32_t)src[2 * n + i] << 16) | ((uint32_t)src[3 * n + i] << 24);
            if ((((uintptr_t)dst) & 3) == 0) ((uint32_t*)dst)[i] = v;
            else { dst[4 * i] = (uint8_t)v; dst[4 * i + 1] = (uint8_t)(v >> 8); dst[4 * i + 2] = (uint8_t)(v >> 16); dst[4 * i + 3] = (uint8_t)(v >> 24); }
        } else if (ts == 8 && (((uintptr_t)dst) & 7) == 0) {
            uint64_t v = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) v |= (uint64_t)src[(int64_t)j * n + i] << (8 * j);
            ((uint64_t*)dst)[i] = v;
        } else {
            for (int j = 0; j < ts; ++j) dst[i * ts + j] = src[(int64_t)j * n + i];
        }
    }
    const int rem = b.bsize % ts;
    if (blockIdx.x == 0 && (int)threadIdx.x < rem) dst[b.bsize - rem + threadIdx.x] = src[b.bsize - rem + threadIdx.x];
}

}  // namespace afhip
