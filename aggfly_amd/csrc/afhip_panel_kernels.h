// afhip_panel_kernels.h — the small kernels after the streaming pass: slot merge + shared
// validity, CSR weighted sums, divide.  Included by afhip_api.hip only (non-template
// __global__ functions must live in one translation unit).
#pragma once
#include "afhip_kernels.h"

namespace afhip {

// ---------------------------------------------------------------------------------------
// k_combine_slots: partial[slot][K][C] -> cell values, validity, cell-major panel.
//
// One thread per (period p, cell c).  For each column the slots of p are merged in slot
// (= time) order; OUT_MEAN divides by the period's inner-group count.  A period without
// slots is an empty resample bin -> NaN (nb_kernels.py:138-141).
//   cells_out [K][P][C]   (optional) per-cell values, NaN kept       = aggregate_time output
//   panel     [C][P][K+1]  where(valid, x, 0) per column and the valid flag in slot K: a thread
//             writes its (K+1) values contiguously, and the CSR kernel reads whole rows
// ---------------------------------------------------------------------------------------
struct CombineArgs {
    const double* partial;
    const int32_t* slot_ptr;       // device [P+1]: slots of period p = [slot_ptr[p], slot_ptr[p+1])
    const int64_t* outer_bounds;   // device [P+1]
    double* cells_out;             // may be null
    double* panel;                 // may be null
    int64_t C, P;
    int32_t K;
    int32_t outer[MAX_COLS];
    int32_t round_final[MAX_COLS];   // 1: round the merged value to float32 (reference dtype rule)
    PackFmt pk;                      // pk.nw != 0: partial holds packed counts, see FusedArgs::packed
};

// partial value of (slot s, column j, cell c) in either layout
__device__ __forceinline__ double ld_partial(const double* partial, const PackFmt& pk, int64_t s, int j, int K, int64_t C, int64_t c) {
    if (pk.nw) {
        const uint64_t q = ((const uint64_t*)partial)[(s * C + c) * pk.nw + pk.word[j]];
        const uint32_t u = (uint32_t)(q >> pk.shift[j]) & pk.mask;
        return u == pk.mask ? nan64() : (double)u;
    }
    return partial[(s * K + j) * C + c];
}

// one packed record (2 or 4 words) of (slot s, cell c) in registers, and a field of it
__device__ __forceinline__ void ld_record(const void* partial, const PackFmt& pk, int64_t s, int64_t C, int64_t c, uint64_t (&q)[4]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const char* rec = (const char*)partial + (s * C + c) * (int64_t)(pk.nw * 8);
    const u4 lo = *(const u4*)rec;
    q[0] = (uint64_t)lo.x | ((uint64_t)lo.y << 32); q[1] = (uint64_t)lo.z | ((uint64_t)lo.w << 32);
    q[2] = 0ull; q[3] = 0ull;
    if (pk.nw == 4) {
        const u4 hi = *(const u4*)(rec + 16);
        q[2] = (uint64_t)hi.x | ((uint64_t)hi.y << 32); q[3] = (uint64_t)hi.z | ((uint64_t)hi.w << 32);
    }
}
__device__ __forceinline__ uint32_t record_field(const uint64_t (&q)[4], const PackFmt& pk, int j) {
    const int wi = pk.word[j];
    const uint64_t w = wi == 0 ? q[0] : (wi == 1 ? q[1] : (wi == 2 ? q[2] : q[3]));
    return (uint32_t)(w >> pk.shift[j]) & pk.mask;
}

__global__ __launch_bounds__(WG) void k_combine_slots(const CombineArgs a) {
    const int64_t c = (int64_t)blockIdx.x * WG + threadIdx.x;   // grid = (cell tiles, period lanes)
    if (c >= a.C) return;
    const int K = a.K;
    const int64_t Q = (int64_t)(K + 1) * a.P;
    for (int64_t p = blockIdx.y; p < a.P; p += gridDim.y) {
        const int s0 = a.slot_ptr[p], s1 = a.slot_ptr[p + 1];
        const double ng = (double)(a.outer_bounds[p + 1] - a.outer_bounds[p]);
        bool valid = true;
        for (int j = 0; j < K; ++j) {
            double v;
            if (s1 == s0) {
                v = nan64();
            } else {
                v = ld_partial(a.partial, a.pk, s0, j, K, a.C, c);
                const int o = a.outer[j];
                for (int s = s0 + 1; s < s1; ++s) {
                    const double x = ld_partial(a.partial, a.pk, s, j, K, a.C, c);
                    if (o == OUT_MIN) { double t = (x < v) ? x : v; v = (x != x || v != v) ? nan64() : t; }
                    else if (o == OUT_MAX) { double t = (x > v) ? x : v; v = (x != x || v != v) ? nan64() : t; }
                    else if (o == OUT_FIRST) { /* a period is never split for OUT_FIRST */ }
                    else v += x;
                }
                if (o == OUT_MEAN) v = v / ng;
                if (a.round_final[j]) v = (double)(float)v;
            }
            valid = valid && (v == v);
            if (a.cells_out) a.cells_out[((int64_t)j * a.P + p) * a.C + c] = v;
            if (a.panel) a.panel[c * Q + p * (K + 1) + j] = v;   // zeroed below if invalid
        }
        if (a.panel) {
            if (!valid)
                for (int j = 0; j < K; ++j) a.panel[c * Q + p * (K + 1) + j] = 0.0;
            a.panel[c * Q + p * (K + 1) + K] = valid ? 1.0 : 0.0;
        }
    }
}

// Tiled form for many output periods: a block owns 32 cells x 8 consecutive periods.  Phase 1
// (thread = one (period, cell)) merges the slots exactly like k_combine_slots and parks the
// K+1 values in LDS; phase 2 writes each cell's 8*(K+1) contiguous doubles of the panel with
// consecutive lanes on consecutive addresses.  The one-thread-per-(p,c) kernel scatters 8-byte
// stores 28 KB apart when P is large (2.7 ms at 251 periods x 13 columns x 51,840 cells).
constexpr int CT_CELLS = 32, CT_PER = 8;

__global__ __launch_bounds__(WG) void k_combine_slots_tiled(const CombineArgs a) {
    __shared__ double tile[CT_CELLS][CT_PER][MAX_COLS + 1];
    const int cl = threadIdx.x % CT_CELLS, pl = threadIdx.x / CT_CELLS;
    const int64_t cbase = (int64_t)blockIdx.x * CT_CELLS;     // grid = (cell tiles, period tiles)
    const int64_t c = cbase + cl;
    const int64_t p0 = (int64_t)blockIdx.y * CT_PER;
    const int64_t p = p0 + pl;
    const int K = a.K;
    const int64_t Q = (int64_t)(K + 1) * a.P;
    if (c < a.C && p < a.P) {
        const int s0 = a.slot_ptr[p], s1 = a.slot_ptr[p + 1];
        const double ng = (double)(a.outer_bounds[p + 1] - a.outer_bounds[p]);
        bool valid = true;
        double vals[MAX_COLS];
        if (a.pk.nw && s1 == s0 + 1) {
            // packed counts: the cell's record in one or two 16-byte loads
            uint64_t q[4];
            ld_record(a.partial, a.pk, s0, a.C, c, q);
#pragma unroll
            for (int j = 0; j < MAX_COLS; ++j) {
                if (j < K) {
                    const uint32_t u = record_field(q, a.pk, j);
                    vals[j] = (u == a.pk.mask) ? nan64() : (double)u;
                } else {
                    vals[j] = 0.0;
                }
            }
        } else if (s1 == s0 + 1) {
            // one slot per period (single-level plans, packed periods): K independent loads in flight
            // instead of a load -> use chain per column (the kernel was latency-bound: 0.95 ms for 2.8 GB)
#pragma unroll
            for (int j = 0; j < MAX_COLS; ++j) vals[j] = (j < K) ? a.partial[((int64_t)s0 * K + j) * a.C + c] : 0.0;
#pragma unroll
            for (int j = 0; j < MAX_COLS; ++j)
                if (j < K && a.outer[j] == OUT_MEAN) vals[j] = vals[j] / ng;
        } else {
#pragma unroll
            for (int j = 0; j < MAX_COLS; ++j) {
                if (j >= K) continue;
                double v;
                if (s1 == s0) {
                    v = nan64();
                } else {
                    v = ld_partial(a.partial, a.pk, s0, j, K, a.C, c);
                    const int o = a.outer[j];
                    for (int s = s0 + 1; s < s1; ++s) {
                        const double x = ld_partial(a.partial, a.pk, s, j, K, a.C, c);
                        if (o == OUT_MIN) { double t = (x < v) ? x : v; v = (x != x || v != v) ? nan64() : t; }
                        else if (o == OUT_MAX) { double t = (x > v) ? x : v; v = (x != x || v != v) ? nan64() : t; }
                        else if (o == OUT_FIRST) { }
                        else v += x;
                    }
                    if (o == OUT_MEAN) v = v / ng;
                }
                vals[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < MAX_COLS; ++j) {
            if (j >= K) continue;
            double v = vals[j];
            if (s1 != s0 && a.round_final[j]) v = (double)(float)v;
            valid = valid && (v == v);
            if (a.cells_out) a.cells_out[((int64_t)j * a.P + p) * a.C + c] = v;
            tile[cl][pl][j] = v;
        }
        if (!valid)
            for (int j = 0; j < K; ++j) tile[cl][pl][j] = 0.0;
        tile[cl][pl][K] = valid ? 1.0 : 0.0;
    }
    __syncthreads();
    if (!a.panel) return;
    const int np = (int)((a.P - p0) < CT_PER ? (a.P - p0) : CT_PER);       // periods in this tile
    const int run = np * (K + 1);                                           // contiguous doubles per cell
    const int ncell = (int)((a.C - cbase) < CT_CELLS ? (a.C - cbase) : CT_CELLS);
    for (int e = threadIdx.x; e < ncell * run; e += WG) {
        const int cc = e / run, off = e - cc * run;
        const int pp = off / (K + 1), jj = off - pp * (K + 1);
        a.panel[(cbase + cc) * Q + p0 * (K + 1) + off] = tile[cc][pp][jj];
    }
}

// Standalone reducers: slots -> the reference's out[G, cell, D_out] layout and dtype
// (nb_kernels.py:257-268: float64 accumulate, store in the input dtype); a pass holds the D columns [d_off, d_off + D) of it.
template <typename TOut>
__global__ __launch_bounds__(WG) void k_slots_to_block(const double* partial, const int32_t* slot_ptr,
                                                       TOut* out, int64_t C, int64_t G, int D, int D_out, int d_off, const PackFmt pk) {
    const int64_t c = (int64_t)blockIdx.x * WG + threadIdx.x;   // grid = (cell tiles, group lanes)
    if (c >= C) return;
    for (int64_t g = blockIdx.y; g < G; g += gridDim.y) {
        const int s0 = slot_ptr[g], s1 = slot_ptr[g + 1];
        for (int d = 0; d < D; ++d) {
            const double v = (s1 == s0) ? nan64() : ld_partial(partial, pk, s0, d, D, C, c);
            out[(g * C + c) * D_out + d_off + d] = (TOut)v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_csr_spmm_counts: the weighted sums of a bin-count plan straight from the packed counts
// (FusedArgs::packed: one 16- or 32-byte record per (slot, cell), all ones = NaN), without the cell-major panel in
// between.  One thread per (region r, period p): per table entry it reads the cell's record once (one or two 16-byte
// loads, four entries in flight) and feeds its K + 1 running sums; the thread's K + 1 results are contiguous in out.
// Column K is the validity weight sum (den).  Same products, same order, same rounding as k_combine_slots +
// k_csr_spmm: x = where(valid, count, 0), valid = no NaN among the K columns; a count plan's columns are NaN
// together (empty period) or not at all, so column 0 tells.
// ---------------------------------------------------------------------------------------
// Rows may be segments of regions (launch_spmm in afhip_api.hip): row v = entries [indptr[v], indptr[v+1]); its record goes
// to row dst_row[v] of `out` (null: v itself).
__global__ __launch_bounds__(WG) void k_csr_spmm_counts(const int64_t* __restrict__ indptr, const int32_t* __restrict__ dst_row,
                                                        const int32_t* __restrict__ cols,
                                                        const double* __restrict__ w, const void* __restrict__ packed,
                                                        const int32_t* __restrict__ slot_ptr, double* __restrict__ out,
                                                        int64_t R, int64_t P, int K, int64_t C, const PackFmt pk,
                                                        double* __restrict__ num = nullptr, double* __restrict__ den = nullptr,
                                                        double* __restrict__ res = nullptr) {
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;       // = r * P + p: a wave walks the periods of one region
    if (tid >= R * P) return;
    const int64_t r = tid / P, p = tid - r * P;
    double* dst = out + ((dst_row ? (int64_t)dst_row[r] : r) * P + p) * (K + 1);
    const int s0 = slot_ptr[p], s1 = slot_ptr[p + 1];
    double acc[MAX_COLS + 1];
#pragma unroll
    for (int k = 0; k <= MAX_COLS; ++k) acc[k] = 0.0;
    if (s1 != s0) {                                                    // else: empty resample bin, every cell invalid
        const int64_t j0 = indptr[r], j1 = indptr[r + 1];
        auto feed = [&](double wj, const uint64_t (&q)[4]) {
            const bool valid = record_field(q, pk, 0) != pk.mask;
#pragma unroll
            for (int k = 0; k < MAX_COLS; ++k) {
                if (k < K) {
                    const double x = valid ? (double)record_field(q, pk, k) : 0.0;
                    acc[k] = __dadd_rn(acc[k], __dmul_rn(wj, x));
                }
            }
            acc[MAX_COLS] = __dadd_rn(acc[MAX_COLS], __dmul_rn(wj, valid ? 1.0 : 0.0));
        };
        int64_t j = j0;
        for (; j + 4 <= j1; j += 4) {          // (eight in flight measured no better: 0.30 vs 0.27 ms on configs[3])
            uint64_t q[4][4];
            double wj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ld_record(packed, pk, s0, C, cols[j + u], q[u]); wj[u] = w[j + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) feed(wj[u], q[u]);
        }
        for (; j < j1; ++j) {
            uint64_t q[4];
            ld_record(packed, pk, s0, C, cols[j], q);
            feed(w[j], q);
        }
    }
    if (res != nullptr) {
        // no row of this table is cut into segments (the host checked): the thread holds the K + 1 sums of its (region, period) and
        // finishes the panel itself — k_panel_divide's arithmetic (spatial.py:127-133) without writing the sums and reading them back
        const int64_t RP = R * P, rp = r * P + p;
        const double de = acc[MAX_COLS];
        if (den) den[rp] = de;
#pragma unroll
        for (int k = 0; k < MAX_COLS; ++k) {
            if (k >= K) continue;
            if (num) num[(int64_t)k * RP + rp] = acc[k];
            res[(int64_t)k * RP + rp] = (de != 0.0) ? acc[k] / de : nan64();
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < MAX_COLS; ++k)
        if (k < K) dst[k] = acc[k];
    dst[K] = acc[MAX_COLS];
}

// k_csr_spmm_counts_sub: the same sums with SUB lanes per (row, period) pair, pairs dealt period-major.  k_csr_spmm_counts gives a
// pair to ONE lane, whose 16-byte reads lie a slot (C records) from its neighbour lanes' — a 64-byte memory transaction per record
// (configs[3]: 208 MB of records gathered at 0.94 TB/s, 0.22 of a 3.1 ms step).  Here the lanes of a group take the row's entries in
// turn — neighbouring cells of one slot: neighbouring records — and the groups of a wave are neighbouring rows of one period; the
// group's K + 1 partial sums are added by DPP (row_shr:1, 2, 4[, 8]: the group's last lane ends with the total).  The sums are a fixed
// tree over the row's entries instead of the table's order (<= 1e-15 apart; `exact_order` plans keep k_csr_spmm_counts).
template <int SUB>
__global__ __launch_bounds__(WG) void k_csr_spmm_counts_sub(const int64_t* __restrict__ indptr, const int32_t* __restrict__ dst_row,
                                                            const int32_t* __restrict__ cols,
                                                            const double* __restrict__ w, const void* __restrict__ packed,
                                                            const int32_t* __restrict__ slot_ptr, double* __restrict__ out,
                                                            int64_t R, int64_t P, int K, int64_t C, const PackFmt pk,
                                                            double* __restrict__ num, double* __restrict__ den, double* __restrict__ res) {
    static_assert(SUB == 4 || SUB == 8 || SUB == 16, "a power-of-two share of a row of 16 lanes");
    const int64_t n = R * P;
    const int64_t first = ((int64_t)blockIdx.x * WG + (threadIdx.x & ~63)) / SUB;     // the wave's first pair
    if (first >= n) return;                                            // whole waves leave together
    const int64_t gid = ((int64_t)blockIdx.x * WG + threadIdx.x) / SUB;
    const bool live = gid < n;                                         // (a last wave's spare groups redo the last pair, unstored)
    const int64_t g = live ? gid : n - 1;
    const int64_t p = g / R, r = g - p * R;                            // period-major: a wave = neighbouring rows of one period
    const int sl = threadIdx.x & (SUB - 1);
    const int s0 = slot_ptr[p], s1 = slot_ptr[p + 1];
    double acc[MAX_COLS + 1];
#pragma unroll
    for (int k = 0; k <= MAX_COLS; ++k) acc[k] = 0.0;
    if (s1 != s0) {                                                    // else: empty resample bin, every cell invalid
        const int64_t j1 = indptr[r + 1];
        for (int64_t j = indptr[r] + sl; j < j1; j += SUB) {
            uint64_t q[4];
            ld_record(packed, pk, s0, C, cols[j], q);
            const double wj = w[j];
            const bool valid = record_field(q, pk, 0) != pk.mask;
#pragma unroll
            for (int k = 0; k < MAX_COLS; ++k) {
                if (k < K) {
                    const double x = valid ? (double)record_field(q, pk, k) : 0.0;
                    acc[k] = __dadd_rn(acc[k], __dmul_rn(wj, x));
                }
            }
            acc[MAX_COLS] = __dadd_rn(acc[MAX_COLS], __dmul_rn(wj, valid ? 1.0 : 0.0));
        }
    }
    // the group's lanes -> its last lane: acc += acc of the lane 1, 2, 4 (, 8) to the left (a group never straddles a row of 16 lanes)
    auto shr = [&](double x, int st) -> double {
        const int xl = __double2loint(x), xh = __double2hiint(x);
        int lo, hi;
        switch (st) {
            case 0: lo = __builtin_amdgcn_update_dpp(0, xl, 0x111, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, xh, 0x111, 0xf, 0xf, true); break;
            case 1: lo = __builtin_amdgcn_update_dpp(0, xl, 0x112, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, xh, 0x112, 0xf, 0xf, true); break;
            case 2: lo = __builtin_amdgcn_update_dpp(0, xl, 0x114, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, xh, 0x114, 0xf, 0xf, true); break;
            default: lo = __builtin_amdgcn_update_dpp(0, xl, 0x118, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, xh, 0x118, 0xf, 0xf, true); break;
        }
        return __hiloint2double(hi, lo);
    };
#pragma unroll
    for (int st = 0; (1 << st) < SUB; ++st) {
#pragma unroll
        for (int k = 0; k <= MAX_COLS; ++k)
            if (k < K || k == MAX_COLS) acc[k] = __dadd_rn(acc[k], shr(acc[k], st));
    }
    if (sl != SUB - 1 || !live) return;
    if (res != nullptr) {
        const int64_t RP = R * P, rp = r * P + p;
        const double de = acc[MAX_COLS];
        if (den) den[rp] = de;
#pragma unroll
        for (int k = 0; k < MAX_COLS; ++k) {
            if (k >= K) continue;
            if (num) num[(int64_t)k * RP + rp] = acc[k];
            res[(int64_t)k * RP + rp] = (de != 0.0) ? acc[k] / de : nan64();
        }
        return;
    }
    double* dst = out + ((dst_row ? (int64_t)dst_row[r] : r) * P + p) * (K + 1);
#pragma unroll
    for (int k = 0; k < MAX_COLS; ++k)
        if (k < K) dst[k] = acc[k];
    dst[K] = acc[MAX_COLS];
}

// ---------------------------------------------------------------------------------------
// k_csr_spmm: out[dst[v]][q] = sum_j w[j] * X[col[j]][q], j over row v = [indptr[v], indptr[v+1]) in table order (dst null:
// v itself; rows may be segments of regions, see launch_spmm in afhip_api.hip).
// One thread per (v, q), q fastest, so a wave reads whole rows of X contiguously.  The
// product is rounded before the add (no FMA): np.add.at adds the already-rounded
// contrib = w * block[...] (spatial.py:183-185).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_csr_spmm(const int64_t* __restrict__ indptr, const int32_t* __restrict__ dst,
                                                 const int32_t* __restrict__ cols,
                                                 const double* __restrict__ w,
                                                 const double* __restrict__ X, double* __restrict__ out,
                                                 int64_t R, int64_t Q) {
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (tid >= R * Q) return;
    const int64_t r = tid / Q, q = tid - r * Q;
    const int64_t j0 = indptr[r], j1 = indptr[r + 1];
    double accv = 0.0;
    int64_t j = j0;
    // eight independent gathers in flight, then the adds in table order
    for (; j + 8 <= j1; j += 8) {
        double p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = __dmul_rn(w[j + u], X[(int64_t)cols[j + u] * Q + q]);
#pragma unroll
        for (int u = 0; u < 8; ++u) accv = __dadd_rn(accv, p[u]);
    }
    for (; j < j1; ++j) accv = __dadd_rn(accv, __dmul_rn(w[j], X[(int64_t)cols[j] * Q + q]));
    out[(dst ? (int64_t)dst[r] : r) * Q + q] = accv;
}

// ---------------------------------------------------------------------------------------
// k_csr_spmm_wave: the same sums for few output columns (Q <= QB <= 16: annual panels, K + 1 values per cell), one WAVE per
// row (segment).  One thread per (row, q) leaves a wave with 64 / Q different rows, each lane walking its own row alone:
// uncoalesced index / weight reads, and the longest of those rows holds the other lanes.  Here the lanes stride over the
// row's entries (cols / w read as contiguous runs, 64 x UNR gathers in flight per wave), every lane keeps Q partial sums,
// and a fixed xor butterfly adds the 64 partials: deterministic, the same for any launch shape, but not the table
// order of np.add.at (spatial.py:185) — plans created with exact_order take k_csr_spmm instead.
// ---------------------------------------------------------------------------------------
template <int QB>
__global__ __launch_bounds__(WG) void k_csr_spmm_wave(const int64_t* __restrict__ seg_ptr, const int32_t* __restrict__ dst,
                                                      const int32_t* __restrict__ cols, const double* __restrict__ w,
                                                      const double* __restrict__ X, double* __restrict__ out,
                                                      int64_t nseg, int Q) {
    const int64_t v = (int64_t)blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (v >= nseg) return;                                        // whole waves leave together: no partial-wave shuffles
    const int lane = threadIdx.x & 63;
    const int64_t j0 = seg_ptr[v], j1 = seg_ptr[v + 1];
    double acc[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) acc[q] = 0.0;
    constexpr int UNR = QB <= 4 ? 4 : 2;                          // entries in flight per lane
    int64_t j = j0 + lane;
    for (; j + (UNR - 1) * 64 < j1; j += UNR * 64) {
        double x[UNR][QB], wj[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t c = (int64_t)cols[j + u * 64] * Q;
            wj[u] = w[j + u * 64];
#pragma unroll
            for (int q = 0; q < QB; ++q) x[u][q] = (q < Q) ? X[c + q] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[q] = __dadd_rn(acc[q], __dmul_rn(wj[u], x[u][q]));
    }
    for (; j < j1; j += 64) {
        const int64_t c = (int64_t)cols[j] * Q;
        const double wj = w[j];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = __dadd_rn(acc[q], __dmul_rn(wj, (q < Q) ? X[c + q] : 0.0));
    }
    // xor butterfly over the 64 lanes: every lane ends with the same sum, the shape never depends on the data
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = __dadd_rn(acc[q], __shfl_xor(acc[q], m, 64));
    }
    if (lane == 0) {
        double* o = out + (int64_t)dst[v] * Q;
#pragma unroll
        for (int q = 0; q < QB; ++q)
            if (q < Q) o[q] = acc[q];
    }
}

// ---------------------------------------------------------------------------------------
// k_csr_spmm_slots: k_combine_slots + k_csr_spmm_wave in ONE pass, straight from partial[slot][K][C] — the cell-major panel
// (K + 1 doubles per cell and period, written once and read back once: 0.5 GB on the reference's own benchmark shape, 12
// monthly periods x 4 columns x 1.04 M cells, where the two kernels took 0.48 ms beside a 5.8 ms streaming pass) is never
// built.  A group of SUB lanes owns one (row segment v, period p): the lanes stride over the segment's table entries (cols /
// w read as contiguous runs; for regions that are runs of neighbouring cells, so are the K plane reads), every lane merges
// the period's slots of its cell exactly like k_combine_slots (slot order; OUT_MEAN / ng; the reference's float32 rounding),
// applies the shared validity rule (spatial.py:114-123: x = where(valid, x, 0), valid = no NaN among the K columns) and keeps
// K + 1 partial sums; a fixed xor butterfly adds the SUB partials.  SUB = 64 reproduces k_csr_spmm_wave's sums bit for bit
// (same per-lane order, same butterfly); smaller groups serve tables whose rows hold a handful of cells (several (v, p) pairs
// per wave instead of 56 idle lanes).  SUB depends on the table only, never on the data: deterministic.  Not the table order
// of np.add.at — exact_order plans keep k_combine_slots + k_csr_spmm.
// ---------------------------------------------------------------------------------------
struct SlotSpmmArgs {
    const int64_t* seg_ptr;          // device [nseg + 1]
    const int32_t* dst;              // device [nseg]: row of `out` a segment's sums go to
    const int32_t* cols;
    const double* w;
    const double* partial;           // device [n_slots][K][C]
    const int32_t* slot_ptr;         // device [P + 1]
    const int64_t* outer_bounds;     // device [P + 1]
    double* out;                     // device [rows][P][K + 1]
    int64_t nseg, P, C;
    int32_t K;
    int32_t p_major;                 // order of the (segment, period) pairs over the grid: 0 = a segment's periods side by side (its table entries
                                     // stay in cache across them), 1 = a period's segments side by side (neighbouring regions share the
                                     // cache lines of ONE slot's planes: with many periods the planes of 365 slots do not fit any cache)
    int32_t outer[MAX_COLS];
    int32_t round_final[MAX_COLS];
    // res != null (no row of the table is cut into segments: segment v IS region dst[v]): the lane that holds the K + 1 sums writes
    // num[K][R][P], den[R][P], res[K][R][P] itself (k_panel_divide's arithmetic, spatial.py:127-133)
    double* num;
    double* den;
    double* res;
    int64_t R;
};

template <int KB, int SUB>
__global__ __launch_bounds__(WG) void k_csr_spmm_slots(const SlotSpmmArgs a) {
    static_assert(SUB == 8 || SUB == 16 || SUB == 32 || SUB == 64, "a power-of-two share of a wave");
    const int64_t n = a.nseg * a.P;
    const int64_t first = ((int64_t)blockIdx.x * WG + (threadIdx.x & ~63)) / SUB;     // the wave's first (v, p) pair
    if (first >= n) return;                                        // whole waves leave together
    const int64_t gid = ((int64_t)blockIdx.x * WG + threadIdx.x) / SUB;
    const bool live = gid < n;                                     // (a last wave's spare groups redo the last pair, unstored)
    const int64_t g = live ? gid : n - 1;
    const int64_t v = a.p_major ? g % a.nseg : g / a.P, p = a.p_major ? g / a.nseg : g - v * a.P;
    const int sl = threadIdx.x & (SUB - 1);
    const int K = a.K;
    const int64_t C = a.C;
    const int s0 = a.slot_ptr[p], s1 = a.slot_ptr[p + 1];
    const double ng = (double)(a.outer_bounds[p + 1] - a.outer_bounds[p]);
    double acc[KB + 1];
#pragma unroll
    for (int k = 0; k <= KB; ++k) acc[k] = 0.0;
    constexpr int UNR = KB <= 4 ? 4 : (KB <= 8 ? 2 : 1);           // entries in flight per lane
    auto finish = [&](double (&x)[KB], double wj) {                // mean / rounding / validity / the K + 1 products
        bool valid = true;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (k < K) {
                if (a.outer[k] == OUT_MEAN) x[k] = x[k] / ng;
                if (a.round_final[k]) x[k] = (double)(float)x[k];
                valid = valid && (x[k] == x[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k)
            if (k < K) acc[k] = __dadd_rn(acc[k], __dmul_rn(wj, valid ? x[k] : 0.0));
        acc[KB] = __dadd_rn(acc[KB], __dmul_rn(wj, valid ? 1.0 : 0.0));
    };
    auto merged = [&](int k, int64_t c) -> double {                // a period cut into several slots: k_combine_slots' merge
        double val = a.partial[((int64_t)s0 * K + k) * C + c];
        const int o = a.outer[k];
        for (int s = s0 + 1; s < s1; ++s) {
            const double x = a.partial[((int64_t)s * K + k) * C + c];
            if (o == OUT_MIN) { double t = (x < val) ? x : val; val = (x != x || val != val) ? nan64() : t; }
            else if (o == OUT_MAX) { double t = (x > val) ? x : val; val = (x != x || val != val) ? nan64() : t; }
            else if (o == OUT_FIRST) { /* a period is never split for OUT_FIRST */ }
            else val += x;
        }
        return val;
    };
    if (s1 != s0) {                                                // else: empty resample bin — every cell invalid, all sums 0
        const int64_t j0 = a.seg_ptr[v], j1 = a.seg_ptr[v + 1];
        int64_t j = j0 + sl;
        if (s1 == s0 + 1) {
            const double* base = a.partial + (int64_t)s0 * K * C;
            for (; j + (UNR - 1) * SUB < j1; j += UNR * SUB) {
                double x[UNR][KB], wj[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int64_t c = a.cols[j + u * SUB];
                    wj[u] = a.w[j + u * SUB];
#pragma unroll
                    for (int k = 0; k < KB; ++k) x[u][k] = (k < K) ? base[(int64_t)k * C + c] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) finish(x[u], wj[u]);
            }
            for (; j < j1; j += SUB) {
                double x[KB];
                const int64_t c = a.cols[j];
#pragma unroll
                for (int k = 0; k < KB; ++k) x[k] = (k < K) ? base[(int64_t)k * C + c] : 0.0;
                finish(x, a.w[j]);
            }
        } else {
            for (; j < j1; j += SUB) {
                double x[KB];
                const int64_t c = a.cols[j];
#pragma unroll
                for (int k = 0; k < KB; ++k) x[k] = (k < K) ? merged(k, c) : 0.0;
                finish(x, a.w[j]);
            }
        }
    }
    // xor butterfly inside the group: every lane of it ends with the same sums
#pragma unroll
    for (int m = SUB / 2; m >= 1; m >>= 1) {
#pragma unroll
        for (int k = 0; k <= KB; ++k)
            if (k < K || k == KB) acc[k] = __dadd_rn(acc[k], __shfl_xor(acc[k], m, 64));
    }
    if (sl == 0 && live) {
        if (a.res != nullptr) {
            const int64_t RP = a.R * a.P, rp = (int64_t)a.dst[v] * a.P + p;
            const double de = acc[KB];
            if (a.den) a.den[rp] = de;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (k >= K) continue;
                if (a.num) a.num[(int64_t)k * RP + rp] = acc[k];
                a.res[(int64_t)k * RP + rp] = (de != 0.0) ? acc[k] / de : nan64();
            }
            return;
        }
        double* o = a.out + ((int64_t)a.dst[v] * a.P + p) * (K + 1);
#pragma unroll
        for (int k = 0; k < KB; ++k)
            if (k < K) o[k] = acc[k];
        o[K] = acc[KB];
    }
}

// ---------------------------------------------------------------------------------------
// k_rf_reduce: the second half of the region-fused period ends (FusedArgs::rf_w).  The streaming kernel left one weighted sum per
// (slot, run, column) — a run = consecutive cells of one wave tile in one region; this kernel adds a region's runs in run order
// (= cell order): sums[r][p][k] = sum over the runs of region r of rf_out[slot(p)][run][k] (+ the region's "extra" entries: the third,
// fourth ... table entries of cells that sit in more than two regions, weighted here from the values those cells wrote); a period
// without a slot (empty resample bin) gives zeros, i.e. no weight.  One thread per (region, period, column).
// ---------------------------------------------------------------------------------------
// Columns whose outer reducer is the mean (bit k of mean_mask) are divided by the period's inner-group count here: the count is the
// same for every cell of a period, so sum_c w_c (s_c / n) = (sum_c w_c s_c) / n up to the rounding of the division (the per-cell
// routes divide per cell: <= 1 ulp per term apart).
__global__ __launch_bounds__(WG) void k_rf_reduce(const double* __restrict__ rf_out, const int64_t* __restrict__ reg_ptr,
                                                  const int32_t* __restrict__ reg_runs, const int32_t* __restrict__ slot_ptr,
                                                  const int64_t* __restrict__ outer_bounds, uint32_t mean_mask,
                                                  const double* __restrict__ ex, int64_t nx, const int64_t* __restrict__ xreg_ptr,
                                                  const int32_t* __restrict__ xcell, const double* __restrict__ xw,
                                                  double* __restrict__ sums, int64_t R, int64_t P, int K1, int64_t n_runs,
                                                  int64_t slot_stride, int64_t run_stride, int p_major) {
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (tid >= R * P * K1) return;
    const int k = (int)(tid % K1);
    const int64_t rp = tid / K1;
    // slot-major run sums: a period's regions side by side (neighbouring regions' runs are neighbours in ONE slot); run-major: a
    // region's periods side by side (its runs hold their periods contiguously)
    const int64_t r = p_major ? rp % R : rp / P, p = p_major ? rp / R : rp - r * P;
    const int s0 = slot_ptr[p], s1 = slot_ptr[p + 1];
    double acc = 0.0;
    if (s1 != s0) {
        const double* base = rf_out + (int64_t)s0 * slot_stride + k;
        // eight independent gathers in flight, then the adds in run order (one at a time the walk is a chain of memory round trips:
        // 70 us for the 30 runs per region of the reference's benchmark shape)
        int64_t q = reg_ptr[r];
        const int64_t q1 = reg_ptr[r + 1];
        for (; q + 8 <= q1; q += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)reg_runs[q + u] * run_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __dadd_rn(acc, v[u]);
        }
        for (; q < q1; ++q) acc = __dadd_rn(acc, base[(int64_t)reg_runs[q] * run_stride]);
        if (nx) {       // the region's entries on cells that sit in three or more regions (their third, fourth ... entries), in table order
            const double* xb = ex + (int64_t)s0 * nx * K1 + k;
            for (int64_t x = xreg_ptr[r]; x < xreg_ptr[r + 1]; ++x) acc = __dadd_rn(acc, __dmul_rn(xw[x], xb[(int64_t)xcell[x] * K1]));
        }
        if ((mean_mask >> k) & 1u) acc = acc / (double)(outer_bounds[p + 1] - outer_bounds[p]);
    }
    sums[(r * P + p) * K1 + k] = acc;
}

// The pieces of a cut region, added in row order: sums[split_row[i]][q] = sum over scratch rows R + [split_ptr[i], split_ptr[i+1]).
__global__ __launch_bounds__(WG) void k_csr_combine_segments(double* __restrict__ sums, const int32_t* __restrict__ split_row,
                                                             const int32_t* __restrict__ split_ptr, int64_t R, int64_t Q, int64_t n_split) {
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (tid >= n_split * Q) return;
    const int64_t i = tid / Q, q = tid - i * Q;
    double a = 0.0;
    for (int64_t k = split_ptr[i]; k < split_ptr[i + 1]; ++k) a = __dadd_rn(a, sums[(R + k) * Q + q]);
    sums[(int64_t)split_row[i] * Q + q] = a;
}

// sums[R][P][K+1] -> num[K][R][P], den[R][P], res[K][R][P] (spatial.py:127-133).
// One thread per (r, p): a wave reads 64 * (K+1) contiguous doubles and writes K + 1 coalesced
// rows (one thread per output value re-read every 112-byte record K times, 28 bytes apart per lane).
__global__ __launch_bounds__(WG) void k_panel_divide(const double* __restrict__ sums, double* __restrict__ num,
                                                     double* __restrict__ den, double* __restrict__ res,
                                                     int64_t R, int64_t P, int K) {
    const int64_t rp = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t RP = R * P;
    if (rp >= RP) return;
    const double* rec = sums + rp * (K + 1);
    double v[MAX_COLS + 1];
#pragma unroll
    for (int k = 0; k <= MAX_COLS; ++k) v[k] = (k <= K) ? rec[k] : 0.0;      // the whole record in flight at once
    double de = 0.0;
#pragma unroll
    for (int k = 0; k <= MAX_COLS; ++k) if (k == K) de = v[k];
    if (den) den[rp] = de;
#pragma unroll
    for (int k = 0; k < MAX_COLS; ++k) {
        if (k >= K) continue;
        if (num) num[(int64_t)k * RP + rp] = v[k];
        res[(int64_t)k * RP + rp] = (de != 0.0) ? v[k] / de : nan64();
    }
}

// res[k][rp] = num[k][rp] / den[rp] where den != 0 else NaN (spatial.py:127-133) on separate arrays: the divide after the
// all-reduce of a cell-sharded job.
__global__ __launch_bounds__(WG) void k_divide_num_den(const double* __restrict__ num, const double* __restrict__ den,
                                                       double* __restrict__ res, int64_t n, int64_t RP) {
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const double de = den[i % RP];
    res[i] = (de != 0.0) ? num[i] / de : nan64();
}

// x[K][C][nt] -> panel[C][nt][K+1] with shared validity (spatial.py:114-123); used by
// afhip_spatial_wavg, whose input layout is the reference's (cell, time) block per name.
__global__ __launch_bounds__(WG) void k_validity_panel(const double* __restrict__ x, double* __restrict__ panel,
                                                       int64_t C, int64_t nt, int K) {
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (tid >= C * nt) return;
    const int64_t c = tid / nt, t = tid - c * nt;
    const int64_t Q = (int64_t)(K + 1) * nt;
    bool valid = true;
    for (int j = 0; j < K; ++j) { const double v = x[((int64_t)j * C + c) * nt + t]; valid = valid && (v == v); }
    for (int j = 0; j < K; ++j) {
        const double v = x[((int64_t)j * C + c) * nt + t];
        panel[c * Q + t * (K + 1) + j] = valid ? v : 0.0;
    }
    panel[c * Q + t * (K + 1) + K] = valid ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------
// k_transform: the element-wise transforms of the staged path on a whole array — Dataset.power (`np.power(block, exp)`,
// dataset.py:527-543), Dataset.spline's hinge (dataset.py:475-481) and Dataset.interact (`np.multiply`, dataset.py:547-563).
// A plain stream: read x (and other), write out; a thread handles TRANSFORM_PER_THREAD elements a block apart so that
// every load of a wave is one contiguous run.  float64 arithmetic with one rounding at the store, except float32 ->
// float32 hinge / product, which numpy evaluates in float32.
// ---------------------------------------------------------------------------------------
constexpr int TRANSFORM_PER_THREAD = 4;
struct TransformArgs {
    const void* x;
    const void* other;
    void* out;
    int64_t n;
    int32_t x_f32, other_f32, out_f32, tf, iarg, pad;
    double arg;
};

__global__ __launch_bounds__(WG) void k_transform(const TransformArgs a) {
    const int64_t base = (int64_t)blockIdx.x * (WG * TRANSFORM_PER_THREAD) + threadIdx.x;
    double v[TRANSFORM_PER_THREAD], o[TRANSFORM_PER_THREAD];
#pragma unroll
    for (int u = 0; u < TRANSFORM_PER_THREAD; ++u) {
        const int64_t i = base + (int64_t)u * WG;
        const int64_t ic = i < a.n ? i : a.n - 1;                     // clamped: the tail re-reads the last element
        v[u] = a.x_f32 ? (double)((const float*)a.x)[ic] : ((const double*)a.x)[ic];
        o[u] = 1.0;
        if (a.tf == TF_INTER) o[u] = a.other_f32 ? (double)((const float*)a.other)[ic] : ((const double*)a.other)[ic];
    }
    const bool all_f32 = a.x_f32 && a.out_f32;
    if (a.tf == TF_POWI) {
        powi_dd_vec<TRANSFORM_PER_THREAD>(v, a.iarg);
    } else if (a.tf == TF_POW) {
#pragma unroll
        for (int u = 0; u < TRANSFORM_PER_THREAD; ++u) v[u] = pow(v[u], a.arg);
    } else if (a.tf == TF_HINGE) {
        if (all_f32) {
            const float kf = (float)a.arg;
#pragma unroll
            for (int u = 0; u < TRANSFORM_PER_THREAD; ++u) { const float xf = (float)v[u]; v[u] = (double)(((xf > kf) ? 1.0f : 0.0f) * (xf - kf)); }
        } else {
#pragma unroll
            for (int u = 0; u < TRANSFORM_PER_THREAD; ++u) v[u] = ((v[u] > a.arg) ? 1.0 : 0.0) * (v[u] - a.arg);
        }
    } else {                                                            // TF_INTER: float32 x float32 products are exact in float64
#pragma unroll
        for (int u = 0; u < TRANSFORM_PER_THREAD; ++u) v[u] = v[u] * o[u];
    }
#pragma unroll
    for (int u = 0; u < TRANSFORM_PER_THREAD; ++u) {
        const int64_t i = base + (int64_t)u * WG;
        if (i < a.n) {
            if (a.out_f32) ((float*)a.out)[i] = (float)v[u];
            else ((double*)a.out)[i] = v[u];
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_place_box: a decoded chunk [bt][by][bx] (contiguous) -> its box of the time-major cube
// [T][NY][NX] at (t0, y0, x0), only the part [st, st+nt) x [sy, sy+ny) x [sx, sx+nx) of the chunk.
// The ingestion route's scatter (aggfly_amd/io.py): reads are contiguous in x runs, writes are
// nx-element runs; one thread per element, x fastest.  TE = element as stored (2, 4 or 8 bytes).
// ---------------------------------------------------------------------------------------
template <typename TE>
__global__ __launch_bounds__(WG) void k_place_box(const TE* __restrict__ src, TE* __restrict__ dst,
                                                  int64_t by, int64_t bx, int64_t st, int64_t sy, int64_t sx,
                                                  int64_t nt, int64_t ny, int64_t nx,
                                                  int64_t NY, int64_t NX, int64_t t0, int64_t y0, int64_t x0) {
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (i >= nt * ny * nx) return;
    const int64_t x = i % nx, r = i / nx, y = r % ny, t = r / ny;
    dst[((t0 + t) * NY + (y0 + y)) * NX + (x0 + x)] = src[((st + t) * by + (sy + y)) * bx + (sx + x)];
}

// ---------------------------------------------------------------------------------------
// k_read_probe: the temporal kernel's access pattern with no arithmetic — the box's read ceiling for this shape, measured beside
// the kernel it bounds (afhip_read_probe; bench.py: roofline.measured_read_ceiling_GBps).  Every lane owns 8 bytes of a row, a
// single-wave workgroup one column tile, four row loads in flight (non-temporal), integer adds, one dword stored per lane at the
// end: the fastest arm of scripts/probe/read_bw.hip's sweep over load width x depth x workgroup size x chunks x cache policy
// (profiles/r03_read_ceiling.txt: 7.02 TB/s where 256-thread workgroups reach 6.69 and cached loads 6.23).
// ---------------------------------------------------------------------------------------
// Small grids: like the temporal kernel, the launch cuts the time axis into blockIdx.y chunks of `rows_per_chunk` rows so that the card is full.
__global__ __launch_bounds__(64) void k_read_probe(const uint32_t* __restrict__ base, int64_t row_dw, int64_t T_all, int64_t rows_per_chunk, uint32_t* __restrict__ out) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const int64_t lane = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t col = lane * 2;
    if (col >= row_dw) return;                                   // (whole lanes only: row_dw is even)
    const int64_t k0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t T = (T_all - k0) < rows_per_chunk ? (T_all - k0) : rows_per_chunk;
    const uint32_t* p = base + k0 * row_dw + col;
    out += (int64_t)blockIdx.y * (row_dw / 2);
    uint32_t acc = 0;
    int64_t t = 0;
    for (; t + 4 <= T; t += 4) {
        u2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load((const u2*)(p + (int64_t)j * row_dw));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y;
        p += 4 * row_dw;
    }
    for (; t < T; ++t, p += row_dw) { const u2 v = __builtin_nontemporal_load((const u2*)p); acc += v.x + v.y; }
    out[lane] = acc;
}

}  // namespace afhip
