// afhip_lz4_kernels.h — chunk decode in HBM for the ingestion path ("or GPU-side decode").
//
// The reference decodes Zarr chunks on host threads inside its dask graph (aggfly/dataset/dataset.py:697-728; read + decode is
// ~80 % of its end-to-end time, benchmarks/bench_read_scheduler.py:4-8).  Here a Blosc-1 chunk whose streams are LZ4 crosses
// PCIe COMPRESSED; the host only parses the container (block table, stream lengths: afcodec_blosc_lz4_plan in
// aggfly_amd/csrc/blosc1.c) and the GPU does the rest:
//
//   k_lz4_streams      one wave per LZ4 stream (a Blosc block, or one byte plane of a split block; any length).  The stream
//                      is decoded inside LDS: the output goes through a ring of 64 KiB + one 8 KiB segment — LZ4 matches
//                      reach back 65,535 bytes at most — and every segment is written to HBM (16 bytes per lane) as soon as
//                      it is complete; the compressed bytes arrive through a 4 KiB input ring, refilled 1 KiB at a time with
//                      aligned 16-byte loads.  Per sequence ONE LDS read fetches a 64-byte window of the input (a byte per
//                      lane): token, length extensions, the literals (stored straight from the lanes' registers when they
//                      fit the window) and the match offset all come out of it with v_readlane; the match copy is a
//                      lane-parallel LDS -> LDS copy, the source index taken modulo the offset for short offsets
//                      (overlapping matches are periodic).  LDS operations of one wave execute in order, so a copy may
//                      read what the previous instruction wrote.
//   k_unshuffle_blocks Blosc's byte shuffle undone per block: element i's byte j sits at plane j, position i.
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

constexpr int LZ4_SEG = 8192;                      // flush granule of the output ring
constexpr int LZ4_RING_MAX = 65536 + LZ4_SEG;      // history a match may reach + the segment being written
constexpr int LZ4_IN = 4096, LZ4_IN_PIECE = 1024;  // input ring, refill piece
constexpr int LZ4_LDS_MAX = LZ4_RING_MAX + LZ4_IN;

__global__ __launch_bounds__(64) void k_lz4_streams(const uint8_t* __restrict__ comp, const Lz4Stream* __restrict__ streams,
                                                    uint8_t* __restrict__ tmp, uint8_t* __restrict__ out, int32_t* __restrict__ errors,
                                                    int ring) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* const obuf = lds;                                   // [ring]   ring = a multiple of LZ4_SEG, <= LZ4_RING_MAX
    uint8_t* const ibuf = lds + ring;                            // [LZ4_IN]
    const Lz4Stream s = streams[blockIdx.x];
    const int lane = threadIdx.x;
    const uint8_t* __restrict__ src = comp + s.src_off;
    uint8_t* __restrict__ dst = (s.to_out ? out : tmp) + s.dst_off;
    const int csize = s.csize, dsize = s.dsize;
    const bool dst16 = (((uintptr_t)dst) & 15) == 0;
    if (csize == dsize) {                                        // stored stream (Blosc keeps what LZ4 could not shrink)
        if (dst16 && (((uintptr_t)src) & 15) == 0) {
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
    // ---- input ring: stream offset q sits at ibuf[(q + phase) & (LZ4_IN - 1)]: aligned 16-byte pieces of HBM stay aligned ----
    const int phase = (int)((uintptr_t)src & 15);
    const uint8_t* const base16 = src - phase;
    int units = 0;                                               // 16-byte units of [base16, ...) loaded so far
    int p = 0;                                                   // read position in the stream
    auto loaded = [&]() { const int n = units * 16 - phase; return n < csize ? (n < 0 ? 0 : n) : csize; };
    auto refill = [&](int upto) {                                // make stream bytes [p, min(upto, csize)) readable
        if (upto > csize) upto = csize;
        while (loaded() < upto) {
            // the piece about to be overwritten must lie wholly before the read position
            if ((units + LZ4_IN_PIECE / 16) * 16 - LZ4_IN > ((p + phase) & ~15)) break;
            const int u = units + lane;
            if (u * 16 < phase + csize)
                *(uint4*)(ibuf + ((u * 16) & (LZ4_IN - 1))) = *(const uint4*)(base16 + (int64_t)u * 16);
            units += LZ4_IN_PIECE / 16;
        }
        __builtin_amdgcn_s_waitcnt(0);
    };
    auto in_byte = [&](int q) -> int { return ibuf[(q + phase) & (LZ4_IN - 1)]; };
    // ---- output ring ----
    int op = 0, opr = 0;                                         // write position in the stream / in the ring
    auto flush = [&](int from, int n) {                          // stream bytes [from, from + n) -> HBM; from is segment-aligned
        const int r0 = from % ring;                              // (segments never straddle the ring's end)
        if (dst16) {
            const int body = n & ~15;
            for (int i = lane * 16; i < body; i += 64 * 16) *(uint4*)(dst + from + i) = *(const uint4*)(obuf + r0 + i);
            for (int i = body + lane; i < n; i += 64) dst[from + i] = obuf[r0 + i];
        } else {
            for (int i = lane; i < n; i += 64) dst[from + i] = obuf[r0 + i];
        }
    };
    auto advance = [&](int n) {                                  // n bytes were written at op (never across a segment end)
        op += n;
        opr += n;
        if (opr >= ring) opr -= ring;
        if ((op & (LZ4_SEG - 1)) == 0) flush(op - LZ4_SEG, LZ4_SEG);
    };
    auto room = [&]() { return LZ4_SEG - (op & (LZ4_SEG - 1)); };   // bytes up to the end of the segment being written
    bool bad = false;
    // ---- decode: every variable that steers control flow is wave-uniform ----
    while (true) {
        if (p >= csize) { bad = true; break; }
        if (loaded() < (p + 64 < csize ? p + 64 : csize)) refill(p + 2048);
        const int at = p + lane;
        const int w = in_byte(at < csize ? at : csize - 1);      // 64-byte window, one byte per lane
        const int token = __builtin_amdgcn_readlane(w, 0);
        int L = token >> 4, hdr = 1;
        if (L == 15) {                                           // extended literal length: 255, 255, ..., last < 255
            int q = p + 1;
            while (true) {
                if (q >= csize) { bad = true; break; }
                if (loaded() < (q + 64 < csize ? q + 64 : csize)) { p = q; refill(q + 2048); }      // (p only steers the refill)
                const int a2 = q + lane;
                const int w2 = in_byte(a2 < csize ? a2 : csize - 1);
                const uint64_t not255 = __builtin_amdgcn_ballot_w64(w2 != 255 || a2 >= csize);
                const int k = not255 ? __builtin_ctzll(not255) : 64;
                if (k < 64) {
                    if (q + k >= csize) { bad = true; break; }
                    L += 255 * k + __builtin_amdgcn_readlane(w2, k);
                    q += k + 1;
                    break;
                }
                L += 255 * 64;
                q += 64;
                if (L > dsize) { bad = true; break; }
            }
            if (bad) break;
            // literals start at q: fall through with the window invalidated
            p = q;
            hdr = 64;                                            // forces the generic copy and the generic offset read below
            if (L > dsize - op || p + L > csize) { bad = true; break; }
            int left = L;
            while (left > 0) {
                if (loaded() <= p) refill(p + 2048);
                int n = loaded() - p;
                if (n > left) n = left;
                const int rm = room();
                if (n > rm) n = rm;
                for (int i = lane; i < n; i += 64) {
                    int t = opr + i;
                    if (t >= ring) t -= ring;
                    obuf[t] = (uint8_t)in_byte(p + i);
                }
                p += n;
                left -= n;
                advance(n);
            }
        } else {
            if (L > dsize - op || p + 1 + L > csize) { bad = true; break; }
            // short literal run: straight from the window (1 + L <= 16 bytes), split only at a segment end
            const int rm = room();
            const int n1 = L < rm ? L : rm;
            if (lane >= 1 && lane < 1 + n1) {
                int t = opr + lane - 1;
                if (t >= ring) t -= ring;
                obuf[t] = (uint8_t)w;
            }
            if (n1) advance(n1);
            if (L > n1) {
                if (lane >= 1 + n1 && lane < 1 + L) {
                    int t = opr + lane - 1 - n1;
                    if (t >= ring) t -= ring;
                    obuf[t] = (uint8_t)w;
                }
                advance(L - n1);
            }
            p += 1 + L;
        }
        if (p >= csize) break;                                   // the last sequence is literals only
        if (p + 2 > csize) { bad = true; break; }
        int off;
        if (hdr + L + 2 <= 64) {
            off = __builtin_amdgcn_readlane(w, hdr + L) | (__builtin_amdgcn_readlane(w, hdr + L + 1) << 8);
        } else {
            if (loaded() < p + 2) refill(p + 2048);
            const int o2 = in_byte(p + (lane & 1));
            off = __builtin_amdgcn_readlane(o2, 0) | (__builtin_amdgcn_readlane(o2, 1) << 8);
        }
        p += 2;
        int M = (token & 15) + 4;
        if ((token & 15) == 15) {                                // extended match length
            while (true) {
                if (p >= csize) { bad = true; break; }
                if (loaded() < (p + 64 < csize ? p + 64 : csize)) refill(p + 2048);
                const int a2 = p + lane;
                const int w2 = in_byte(a2 < csize ? a2 : csize - 1);
                const uint64_t not255 = __builtin_amdgcn_ballot_w64(w2 != 255 || a2 >= csize);
                const int k = not255 ? __builtin_ctzll(not255) : 64;
                if (k < 64) {
                    if (p + k >= csize) { bad = true; break; }
                    M += 255 * k + __builtin_amdgcn_readlane(w2, k);
                    p += k + 1;
                    break;
                }
                M += 255 * 64;
                p += 64;
                if (M > dsize) { bad = true; break; }
            }
            if (bad) break;
        }
        if (off == 0 || off > op || M > dsize - op) { bad = true; break; }
        // match: byte op + i = byte op - off + i; pieces end at segment ends (a flush may sit between them)
        int left = M;
        while (left > 0) {
            const int rm = room();
            const int n = left < rm ? left : rm;
            int sr = opr - off;                                  // ring index of the piece's first source byte
            if (sr < 0) sr += ring;
            if (off >= 64) {
                // a 64-byte step never reads a byte the same step writes; the ordered LDS queue covers the earlier steps
                for (int i = lane; i < n; i += 64) {
                    int a = sr + i, t = opr + i;
                    if (a >= ring) a -= ring;
                    if (t >= ring) t -= ring;
                    const uint8_t b = obuf[a];
                    obuf[t] = b;
                }
            } else {
                // overlapping match: the last `off` bytes repeat
                for (int i = lane; i < n; i += 64) {
                    int a = sr + (i % off), t = opr + i;
                    if (a >= ring) a -= ring;
                    if (t >= ring) t -= ring;
                    const uint8_t b = obuf[a];
                    obuf[t] = b;
                }
            }
            left -= n;
            advance(n);
        }
    }
    if (bad || op != dsize) {
        if (lane == 0) atomicAdd(errors, 1);
        return;
    }
    if (op & (LZ4_SEG - 1)) flush(op & ~(LZ4_SEG - 1), op & (LZ4_SEG - 1));       // the last, partial segment
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz4_streams_hbm: the same decode WITHOUT the LDS ring — the output is written straight to its destination and a match
// reads its source back from there.  A wave's stores and later loads of the same bytes are ordered by s_waitcnt vmcnt(0)
// (the store has reached L2) and the loads are L2-served (agent-scope relaxed atomics = sc1: the per-CU L1 is not kept
// coherent with the stores).  Every sequence therefore pays one or two L2 round trips — several times the LDS version's
// latency — but the kernel holds no LDS, so 16-32 waves per CU overlap those round trips instead of one or two.
// The input window is read from global memory (read-only, L1-cached).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ld_l2_u8(const uint8_t* p) {
    // 32-bit aligned container load, L2-served; the byte is extracted in registers
    const uint32_t* q = (const uint32_t*)((uintptr_t)p & ~(uintptr_t)3);
    const uint32_t v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)((v >> (8 * ((uintptr_t)p & 3))) & 0xff);
}

// i mod m for 0 <= i < 64 * m (the packed matches of k_lz4_streams_hbm: i < 64): six compare-subtract steps, no division
__device__ __forceinline__ int small_mod(int i, int m) {
#pragma unroll
    for (int sft = 5; sft >= 0; --sft) i -= (i >= (m << sft)) ? (m << sft) : 0;
    return i;
}

constexpr int LZ4_GROUP = 8;       // sequences parsed from one 64-byte window
constexpr int LZ4_NEAR = 4096;     // the last bytes of the output, mirrored in LDS: matches into them wait for nothing

__global__ __launch_bounds__(64) void k_lz4_streams_hbm(const uint8_t* __restrict__ comp, const Lz4Stream* __restrict__ streams,
                                                        uint8_t* tmp, uint8_t* out, int32_t* __restrict__ errors) {
    // Every output byte is also written to near[position mod LZ4_NEAR]: a match whose source lies within the last
    // LZ4_NEAR bytes — the common case on byte planes of smooth fields — reads it from there (LDS operations of a wave run in
    // order: no wait for the store to reach L2, no L2 round trip for the load); older history comes back from HBM / L2.
    __shared__ uint8_t near[LZ4_NEAR];
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
    auto put = [&](int pos, int v) {                              // one output byte: HBM and the near ring
        dst[pos] = (uint8_t)v;
        near[pos & (LZ4_NEAR - 1)] = (uint8_t)v;
    };
    int p = 0, op = 0;
    int acked = 0;                                               // output bytes [0, acked) are known to have reached L2
    bool bad = false;
    int w_next = src[lane < csize ? lane : csize - 1];           // the window of the next iteration, loaded one iteration ahead
    while (true) {
        if (p >= csize) { bad = true; break; }
        const int w = w_next;
        const int wl = (csize - p) < 64 ? (csize - p) : 64;      // valid bytes of the window
        // ---- parse as many whole sequences as the window holds: token (+ at most one extension byte per length), up to 61
        // literals, offset, a match of at most 64 bytes.  One window load then serves up to LZ4_GROUP sequences instead of
        // one; what does not fit takes the generic path below. ----
        uint64_t rec[LZ4_GROUP];                                 // literal start (6 bits) | L (6) << 6 | M (7) << 12 | offset << 19
        int n = 0, c = 0, run = op;
#pragma unroll
        for (int k = 0; k < LZ4_GROUP; ++k) {
            if (n != k || c >= wl) continue;                     // (stopped earlier)
            const int tok = __builtin_amdgcn_readlane(w, c);
            int L = tok >> 4, M = (tok & 15) + 4, pos = c + 1;
            if (L == 15) {
                if (pos >= wl) continue;
                const int e = __builtin_amdgcn_readlane(w, pos);
                if (e == 255) continue;                          // a literal run of 270 or more: generic path
                L += e;
                ++pos;
            }
            const int lc = pos;
            pos += L;
            if (pos + 2 > wl) continue;
            const int off = __builtin_amdgcn_readlane(w, pos) | (__builtin_amdgcn_readlane(w, pos + 1) << 8);
            pos += 2;
            if ((tok & 15) == 15) {
                if (pos >= wl) continue;
                const int e = __builtin_amdgcn_readlane(w, pos);
                M += e;
                ++pos;
                if (e == 255 || M > 64) continue;
            }
            if (p + pos > csize - 1) continue;                   // the stream's last bytes: generic path
            if (L > dsize - run) { bad = true; continue; }
            run += L;
            if (off == 0 || off > run || M > dsize - run) { bad = true; continue; }
            run += M;
            rec[k] = (uint64_t)lc | ((uint64_t)L << 6) | ((uint64_t)M << 12) | ((uint64_t)off << 19);
            c = pos;
            n = k + 1;
        }
        if (bad) break;
        if (n > 0) {
            {   // the next window is on its way while this group's stores and loads run
                const int a2 = p + c + lane;
                w_next = src[a2 < csize ? a2 : csize - 1];
            }
            // ---- all the group's literals in ONE store: a lane of the window knows which sequence its byte belongs to ----
            {
                int my = -1, o = op;
#pragma unroll
                for (int k = 0; k < LZ4_GROUP; ++k) {
                    if (k < n) {
                        const int lc = (int)(rec[k] & 63), L = (int)((rec[k] >> 6) & 63), M = (int)((rec[k] >> 12) & 127);
                        if (lane >= lc && lane < lc + L) my = o + lane - lc;
                        o += L + M;
                    }
                }
                if (my >= 0) put(my, w);
            }
            // ---- matches: consecutive ones that do not read what a pending one writes share the lanes of one load / store
            // pair.  The ring now holds the group's literals too (up to position `run`): bytes from run - LZ4_NEAR on are near. ----
            int my_src = 0, my_dst = -1, used = 0, pend_lo = 0, o = op;
            bool my_near = false;
            auto flush = [&]() {
                if (used) {
                    if (my_dst >= 0) put(my_dst, my_near ? (int)near[my_src & (LZ4_NEAR - 1)] : ld_l2_u8(dst + my_src));
                    my_dst = -1;
                    used = 0;
                }
            };
#pragma unroll
            for (int k = 0; k < LZ4_GROUP; ++k) {
                if (k < n) {
                    const int L = (int)((rec[k] >> 6) & 63), M = (int)((rec[k] >> 12) & 127), off = (int)(rec[k] >> 19);
                    o += L;                                       // the match's destination
                    const int from = o - off, src_end = from + (M < off ? M : off);
                    if (used && (src_end > pend_lo || used + M > 64)) flush();     // reads a pending match's bytes / lanes used up
                    const bool is_near = from >= run - LZ4_NEAR;
                    if (!is_near && src_end > acked) {           // old history that may not have reached L2 yet
                        flush();
                        __builtin_amdgcn_s_waitcnt(0);
                        acked = o;
                    }
                    if (!used) pend_lo = o;
                    const int i = lane - used;
                    if (i >= 0 && i < M) {
                        my_src = from + (off < M ? small_mod(i, off) : i);
                        my_dst = o + i;
                        my_near = is_near;
                    }
                    used += M;
                    o += M;
                }
            }
            flush();
            p += c;
            op = run;
            continue;
        }
        // ---- generic path: one sequence with long length extensions, long copies, or the stream's last sequence ----
        const int token = __builtin_amdgcn_readlane(w, 0);
        int L = token >> 4, hdr = 1;
        bool window_ok = true;                                    // the offset still sits in the window
        if (L == 15) {
            int q = p + 1;
            while (true) {
                if (q >= csize) { bad = true; break; }
                const int a2 = q + lane;
                const int w2 = src[a2 < csize ? a2 : csize - 1];
                const uint64_t not255 = __builtin_amdgcn_ballot_w64(w2 != 255 || a2 >= csize);
                const int k = not255 ? __builtin_ctzll(not255) : 64;
                if (k < 64) {
                    if (q + k >= csize) { bad = true; break; }
                    L += 255 * k + __builtin_amdgcn_readlane(w2, k);
                    q += k + 1;
                    break;
                }
                L += 255 * 64;
                q += 64;
                if (L > dsize) { bad = true; break; }
            }
            if (bad) break;
            hdr = q - p;
            window_ok = false;
        }
        if (L > dsize - op || p + hdr + L > csize) { bad = true; break; }
        if (window_ok && 1 + L <= wl) {
            if (lane >= 1 && lane < 1 + L) put(op + lane - 1, w);
        } else {
            const uint8_t* from = src + p + hdr;
            for (int i = lane; i < L; i += 64) put(op + i, from[i]);
        }
        p += hdr + L;
        op += L;
        if (p >= csize) break;
        if (p + 2 > csize) { bad = true; break; }
        int off;
        if (window_ok && 1 + L + 2 <= wl) {
            off = __builtin_amdgcn_readlane(w, 1 + L) | (__builtin_amdgcn_readlane(w, 2 + L) << 8);
        } else {
            const int o2 = src[p + (lane & 1)];
            off = __builtin_amdgcn_readlane(o2, 0) | (__builtin_amdgcn_readlane(o2, 1) << 8);
        }
        p += 2;
        int M = (token & 15) + 4;
        if ((token & 15) == 15) {
            while (true) {
                if (p >= csize) { bad = true; break; }
                const int a2 = p + lane;
                const int w2 = src[a2 < csize ? a2 : csize - 1];
                const uint64_t not255 = __builtin_amdgcn_ballot_w64(w2 != 255 || a2 >= csize);
                const int k = not255 ? __builtin_ctzll(not255) : 64;
                if (k < 64) {
                    if (p + k >= csize) { bad = true; break; }
                    M += 255 * k + __builtin_amdgcn_readlane(w2, k);
                    p += k + 1;
                    break;
                }
                M += 255 * 64;
                p += 64;
                if (M > dsize) { bad = true; break; }
            }
            if (bad) break;
        }
        if (off == 0 || off > op || M > dsize - op) { bad = true; break; }
        if (off <= LZ4_NEAR - 64) {
            // a near match of any length, 64 bytes a step through the ring.  off >= 64: a step reads bytes of earlier steps /
            // sequences only.  Shorter periods: the first step repeats the last `off` bytes; later steps copy from P bytes
            // back, P = the multiple of `off` in [64, 64 + off): written by earlier steps, never further back than the ring
            // keeps even for matches much longer than the ring
            const int P = off >= 64 ? off : off * ((64 + off - 1) / off);
            for (int i0 = 0; i0 < M; i0 += 64) {
                const int i = i0 + lane;
                if (i < M) put(op + i, near[((i0 == 0 && off < 64) ? op - off + lane % off : op + i - P) & (LZ4_NEAR - 1)]);
            }
        } else {
            // old history: it must have reached L2 — wait for this wave's outstanding stores only when the source touches bytes
            // stored since the last wait
            if (op - off + (M < off ? M : off) > acked) {
                __builtin_amdgcn_s_waitcnt(0);
                acked = op;
            }
            const uint8_t* from = dst + op - off;
            if (off >= M) {                                       // no overlap: all loads first, then the stores
                for (int i = lane; i < M; i += 64) put(op + i, ld_l2_u8(from + i));
            } else {
                // overlapping by whole steps (off > LZ4_NEAR - 64 >= 64): each step's source was stored by an earlier step
                for (int i0 = 0; i0 < M; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < M) put(op + i, ld_l2_u8(from + i));
                    __builtin_amdgcn_s_waitcnt(0);
                }
                acked = op + M;
            }
        }
        op += M;
        {
            const int a2 = p + lane;
            w_next = src[a2 < csize ? a2 : csize - 1];
        }
    }
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
