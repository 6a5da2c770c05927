// afhip_kernels.h — device code of the MI355X (gfx950) aggregation engine.
//
// Three kernels do the work of one aggregate_dataset() call:
//
//   k_fused_temporal  one streaming pass over the raw (time, cell) cube.  A lane owns VEC
//                     neighbouring cells; it walks its time chunk once, keeps every inner
//                     accumulator (sum/min/max, threshold slots) and every column's outer
//                     accumulator in registers, and writes one partial per (slot, column,
//                     cell).  HBM-bound: algorithmic bytes = T * n_cells * sizeof(elem).
//                     Restates _block_stat/_block_dd/_block_bins/_block_sine_dd
//                     (aggfly/aggregate/nb_kernels.py:121-251) and the transforms
//                     (aggfly/dataset/dataset.py:475-481,527-543) per cell, in the
//                     reference's k-ascending order.
//   k_combine_slots   merges the partials of an outer period in slot order, applies the
//                     shared validity rule (aggfly/aggregate/spatial.py:114-119) and lays
//                     the result out cell-major for the sparse stage.
//   k_csr_spmm        region x cell CSR weighted sums in table order
//                     (_scatter_block, spatial.py:181-186), then k_panel_divide
//                     (spatial.py:127-133).
//
// Runs without per-cell output skip the middle kernel: k_csr_spmm_slots gathers the period partials directly (slot merge,
// validity and weighted sums in one pass), and plans with several output periods reduce their cells by region INSIDE
// k_fused_temporal at every period end (FusedArgs::rf_w, rf_emit; k_rf_reduce adds a region's runs) — the per-cell period
// values are then never written.
//
// No MFMA anywhere: the path is a streaming scan plus a sparse segmented sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afhip {

constexpr int WG = 256;          // 4 wavefronts of 64
constexpr int MAX_THR = 16;      // threshold slots evaluated on raw data per pass
constexpr int MAX_COLS = 16;     // output columns per pass
constexpr int HB_TABLE_BYTES = 2 * (MAX_THR + 2) * 16;    // LDS edge tables of the histogram path (576 B)

// inner-source kinds (what a column reads at the end of an inner group)
enum : int { SRC_MEAN = 0, SRC_SUM = 1, SRC_MIN = 2, SRC_MAX = 3, SRC_NANMEAN = 4, SRC_THR = 5, SRC_SINE = 6 };
enum : int { TF_NONE = 0, TF_POWI = 1, TF_POW = 2, TF_HINGE = 3, TF_INTER = 4 };
enum : int { OUT_FIRST = 0, OUT_SUM = 1, OUT_MEAN = 2, OUT_MIN = 3, OUT_MAX = 4, OUT_DD = 5, OUT_BINS = 6 };

// One threshold slot on raw data: contribution = (t0 < v && v < t1) ? fma(A, v, B) : 0.
//   dd, base = t0:  inside the window v - t0 > 0, so |v - base| = fma(+1, v, -t0)
//   dd, base = t1:  inside the window v - t1 < 0, so |v - base| = fma(-1, v, +t1)
//   bins:           fma(0, v, 1) = 1
// One rounding, identical to the reference's av = v - base; if av < 0: av = -av
// (nb_kernels.py:169-177) and c += 1.0 (nb_kernels.py:190-196).
// t0f / t1f are t0 rounded down / t1 rounded up to float: for a float v,
// (double)v > t0  <=>  v > t0f  and  (double)v < t1  <=>  v < t1f, so f32 cubes compare in f32.
// packed-count record format (FusedArgs::packed): nw = 0 -> not packed
struct PackFmt {
    int32_t nw;                      // 64-bit words per (slot, cell): 2 or 4
    uint32_t mask;                   // all ones of a field = NaN
    uint8_t word[MAX_COLS], shift[MAX_COLS];
};

struct ThrSlot {
    double t0, t1, A, B;
    float t0f, t1f;
    int32_t nan_poisons;  // dd: a NaN in the window makes the group NaN; bins: it does not
    int32_t pad;
};

struct ColOp {
    int32_t src, src_idx;      // SRC_*; slot index for SRC_THR
    int32_t tf, tf_iarg;       // TF_*; integer exponent for TF_POWI
    int32_t outer, skind;      // OUT_*; sine_dd kind flag (0 cooling, 1 heating)
    int32_t rounding, inter_f32;   // AFHIP_ROUND_* bits (float32 intermediates like the reference); 1: `inter` holds float32
    const void* inter;         // TF_INTER: the second cube [G1][C] (one value per inner group and cell), else null
    double s0, s1;             // sine_dd thresholds
    double s0x2, s1x2;         // 2 * s0, 2 * s1 (exact): the cooling form's 2 thr - tmax - tmin starts from them
    float s0dn, s0up, s1dn, s1up;  // s0 / s1 rounded down / up to float: for float tmin, tmax   tmin < s <=> tmin < up,  s < tmax <=> tmax > dn
    double swidth, swidth2;        // s1 - s0, 2 (s1 - s0)
    double tf_arg;             // exponent (TF_POW) or knot (TF_HINGE)
    double o0, o1, obase;      // outer dd/bins thresholds
};

struct ChunkDesc {
    int64_t k_lo, k_hi;        // time steps [k_lo, k_hi)
    int32_t g_lo, g_hi;        // inner groups [g_lo, g_hi); k_lo == ib[g_lo], k_hi == ib[g_hi]
    int32_t slot_base, pad;
};

struct FusedArgs {
    const void* cube;
    int64_t C;                     // cells
    const int64_t* gtab;           // device [G1+1][2]: {(end step of inner group g) << 1 | (emit a slot after g),
                                   //                    bits of the double 1.0 / (steps in g)}
    const ChunkDesc* chunks;       // device [n_chunks]
    double* partial;               // device [n_slots][K][C]
    int32_t K, nthr;
    int32_t xcd_remap, n_tiles;    // 1: give each XCD a contiguous range of cell tiles (speed only)
    const double* sine_tab;        // device [2][SINE_ROWS][4]: rows of the acos table (sine_theta), or null when no column is sine_dd
    // LDS-histogram bins (FEAT bit 5): the threshold slots form a contiguous partition of equal
    // width with edges hb_edge[0..hb_n]; hb_bin_of_slot[slot] = position of that slot's bin.
    // guess bin (shifted by one guard bin) = floor(v * hb_c1 + hb_c0); hb_dn / hb_up are the edges
    // rounded down / up to float, so a float v compares exactly:  v > t <=> v > dn,  v < t <=> v < up.
    double hb_c1, hb_c0;
    float hb_c1f, hb_c0f;
    int32_t hb_n, hb_shift;        // hb_shift = log2(blockDim): counters are laid out [bin * VEC + i][blockDim]
    int32_t hb_bin_of_slot[MAX_THR];
    double hb_edge[MAX_THR + 1];
    float hb_dn[MAX_THR + 1], hb_up[MAX_THR + 1];
    // arithmetic edges (FEAT bit 6): every edge is EXACTLY hb_lo0 + g * hb_w in the input precision (host-checked with
    // the same fma), so the two edges around a guess are two fmas instead of an LDS table read; hb_gl / hb_gh sit
    // inside the lower / upper guard bin: values (and NaN) are clamped onto them first.
    // packed != 0 (single-level plans whose columns are all bin counts): the period's K counts leave as small
    // integers, cell-major, one 16- or 32-byte record per (slot, cell), all ones = NaN (empty period) — instead of
    // K doubles per cell in K planes: an eighth / a quarter of the bytes in 1 / 2 stores instead of K (the f64 stores
    // of the 13-bin CMIP6 plan cost its streaming kernel 18 %)
    int32_t packed;
    // record format: pk_nw 64-bit words per (slot, cell) — 2 (16 B) when K counts of pk_bw bits fit, else 4 with 16-bit
    // fields; column j sits in word pk_word[j] at bit pk_shift[j]; all ones (pk_mask) = NaN
    int32_t pk_nw;
    uint32_t pk_mask;
    uint8_t pk_word[MAX_COLS], pk_shift[MAX_COLS];
    double hb_w, hb_lo0, hb_gl, hb_gh;
    float hb_wf, hb_lo0f, hb_glf, hb_ghf;
    // ... and the guess constant biased DOWN (hb_c0 - delta, host-chosen): floor(v * hb_c1 + hb_c0b) is then never above the
    // value's bin and at most one below it, and an edge value E[k] always guesses k - 1 — checked by the host on every edge
    // with the kernel's own fma in the input precision; the fma and the floor are monotone in v, so the edges decide it for
    // every value between them (ha_update)
    double hb_c0b;
    float hb_c0bf;
    int32_t hb_pad;
    ThrSlot thr[MAX_THR];
    ColOp cols[MAX_COLS];
    // the lean group end's view of a column in ONE word (src | tf << 4 | (tf_iarg & 0xff) << 8): six of them stay in scalar
    // registers through the time loop; the full records do not (they were re-read, three scalar loads and their waits, per column
    // and group: 13 s_load per two-row group on a four-column polynomial)
    uint32_t ccode[MAX_COLS];
    // Region-fused period ends (rf_w != null; afhip_api.hip: rf_table): instead of one value per (slot, column, cell) the kernel
    // writes one weighted sum per (slot, RUN, column) — a run = consecutive cells of a wave's tile whose e-th table entry names the
    // same region (e = 0, 1; a cell's third, fourth ... entries are "extras", below).  Per period end a wave multiplies
    // where(valid, x_k, 0) and the valid flag of its cells by the cells' weights and adds each run's products with a segmented scan
    // over its lanes (rf_emit); k_rf_reduce then adds a region's runs in run order.  The per-cell period values — 0.4 GB per
    // launch on the reference's own benchmark shape, stores that cost its streaming kernel 6 % — are never written, and the
    // gather kernel over them does not run.
    const double* rf_w;            // device [C][2]: weight of the cell's first / second table entry (0: none)
    const int32_t* rf_tile;        // device [wave tiles][2][2]: {first run, need mask (rf_need)} of entry e in wave tile t (64 * VEC cells)
    double* rf_out;                // device: the sum of (slot, run, column k) at  slot * rf_slot_stride + run * rf_run_stride + k  —
    int64_t rf_slot_stride, rf_run_stride;      // run-major ([runs][n_slots][K + 1]) for plans of many periods: a region's periods side by side for k_rf_reduce
    // cells that sit in MORE than two regions (junctions of polygons): their third, fourth ... entries are "extras" — such a cell
    // also writes its validated period values to rf_ex[slot][rf_x[cell]][K + 1], and k_rf_reduce adds the extras of a region from there
    const int32_t* rf_x;           // device [C]: index of the cell among the cells with extras, -1 = none; null: the table has none
    double* rf_ex;                 // device [n_slots][rf_nx][K + 1]
    int64_t rf_nx;
    // per lane slot (tile * 64 + lane, tiles padded): {scan / start / end bits, run indices} of its cells for both entries, worked out
    // on the host from the regions of neighbouring cells (afhip_api.hip: rf_table; k_fused_temporal: rfbits / rfrid)
    const uint32_t* rf_lane;       // device [wave tiles * 64][2]
    int32_t rf_lds_off, rf_pad;    // byte offset, in the dynamic LDS, of the waves' parking blocks: RF_LANE_BYTES per lane (weights + lane words)
};

}  // namespace afhip

#include "afhip_numerics.h"
#include "afhip_sine.h"
#include "afhip_loads.h"

namespace afhip {

// ---------------------------------------------------------------------------------------
// k_fused_temporal
//
// grid = (cell tiles, chunks).  blockIdx.x (fastest) walks neighbouring tiles of the same
// chunk, so workgroups resident together read adjacent 4 KB pieces of the same rows: the
// union is one long contiguous run per time step.  There is no reuse between workgroups
// (every byte is read once), so no XCD-aware remap is applied.
//
//   PIPE 0  lanes load straight to registers (any alignment; VEC = 1 is the fallback for
//           shapes whose rows are not 16-byte multiples).
//   PIPE 1  each wave streams its 1 KiB row pieces into a private LDS ring with
//           global_load_lds_dwordx4 (no VGPR destination), DEPTH rows ahead, and reads the
//           current row back with ds_read_b128.  The ring is wave-private, so there are no
//           workgroup barriers: a counted s_waitcnt vmcnt(DEPTH-1) is the only ordering
//           (MI355X_MICROARCH.md, "Two waves per SIMD" item 7).  The prefetch runs across
//           inner-group boundaries, so short groups (daily data, tmin/tmax pairs) stream
//           as well as long ones.
//
//   STAT 0 none | 1 sum | 2 sum+min+max | 3 sum+count+min+max with NaN skipping (nanmean)
// ---------------------------------------------------------------------------------------
//   FEAT bit 0: single-sine degree days compiled in (needs STAT >= 2)
//        bit 1: the general transforms compiled in: pow() with a non-integer exponent, and `inter` (Dataset.interact,
//               dataset.py:483-518,547-563: the inner value times the matching element of a second cube)
// Both are bulky once inlined per column, so only the variants that need them carry them.
//
// The workgroup size is a launch parameter (64 or 256 threads): waves never talk to each
// other, so small grids are launched as single-wave workgroups for a finer tail.
#ifndef AFHIP_RF_DPP
#define AFHIP_RF_DPP 1             // the period end's scan moves its values by DPP (1) or by ds_bpermute (0: the Hillis-Steele steps of round 4's first forms)
#endif
#ifndef AFHIP_RF_WAVES
#define AFHIP_RF_WAVES 1           // waves per SIMD the region-fused twins are compiled for (1 = no constraint)
#endif
template <typename TIn, int PIPE, int VEC, int STAT, int NTHR, int KMAX, int DEPTH, int FEAT>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu((FEAT & 2048) ? AFHIP_RF_WAVES : 1))) void k_fused_temporal(const FusedArgs a) {
    constexpr int AUX = (FEAT & 4) ? 2 : 0;   // FEAT bit 2: non-temporal (nt) cache policy on the streaming loads
    // FEAT bit 3: every threshold slot is a bin count -> 32-bit integer counters (one
    //             v_addc per slot and element instead of fma + select + f64 add)
    // FEAT bit 4: single-level plan (every inner group is an output period, every column
    //             passes its inner value through): no outer accumulators at all
    // FEAT bit 5: the bins are a contiguous equal-width partition -> per-lane histogram in LDS.
    //             One fma + floor in the INPUT precision guesses the bin (off by one at most, host-
    //             checked); the two edges around the guess come from an LDS table and four exact
    //             compares move the guess up / down or reject a value that sits on an edge (strict
    //             inequalities, like the reference); one ds_add_u32 bumps the lane's private
    //             counter.  A guard bin on either side absorbs out-of-range values, so there is no
    //             range test and no data-dependent branch.  ~13 VALU + 2 LDS ops per element
    //             instead of 3 VALU per bin.
    constexpr bool TKI = (FEAT & 8) != 0;
    constexpr bool SL = (FEAT & 16) != 0;
    constexpr bool HB = (FEAT & 32) != 0;
    // FEAT bit 6: histogram with arithmetic edges — exactly representable equal-width edges (5 degC bins from -20 ...):
    //             the edge pair of the guessed bin is computed (2 fma) instead of read from the LDS table, which takes an
    //             LDS round trip out of every element's dependent chain; a value on an edge is recognised by equality.
    constexpr bool HA = (FEAT & 64) != 0;
    // FEAT bit 7: every inner group holds exactly TWO rows ((tmin, tmax) pairs per day, configs[4]): a block of DEPTH rows is
    //             DEPTH / 2 whole groups — all of them in flight at once instead of one group's two rows — and a group's
    //             statistics are min / max / sum of the pair, taken in the input precision, without the per-row accumulators.
    constexpr bool PAIR = (FEAT & 128) != 0;
    // FEAT bit 8: pair mode with the LEAN group end: every column is   mean | sum | min | max | sine_dd  ->  (nothing | integer
    //             power)  ->  sum | mean,   without float32 rounding (configs[4]'s sine_dd -> sum; the daily mean of (tmin, tmax)
    //             and its polynomial).  The group end is then the column's value, its power chain and one add — no per-group walk
    //             through the column records' source / transform / reducer switches (that walk is ~90 scalar instructions per wave
    //             and group, as many as the vector ones that do the arithmetic: profiles/r03_kbench_c5_table_arc.txt); the records
    //             are loop-invariant kernel arguments and stay in scalar registers; a NaN pair is remembered in a lane mask (scalar
    //             OR) and applied when the period's sum leaves the kernel, since NaN is sticky under + anyway.
    constexpr bool LEAN = (FEAT & 256) != 0;
    // FEAT bit 9: ... and every column is a plain sine_dd (no power): nothing but the closed forms in the group end
    constexpr bool LEAN_SINE = (FEAT & 512) != 0;
    static_assert(!LEAN_SINE || (LEAN && (FEAT & 1) && KMAX <= 2), "the sine-only lean form: at most two columns");
    // FEAT bit 10: the same mode for inner groups of exactly FOUR rows (6-hourly data): GL rows per group, DEPTH / GL groups per block;
    //              lean form only.  The sum runs in time order ((u0 + u1) + u2) + u3, the mean is s / 4 = s * 0.25 exactly, min / max
    //              are taken in the input precision; sine_dd columns use the general closed forms (tavg is not the mid-range of four
    //              steps), so these variants read the acos table.
    // FEAT bit 12: ... and of exactly THREE rows (8-hourly data): the sum runs (u0 + u1) + u2, the mean is the correctly rounded s / 3
    //              (div_by with the correctly rounded reciprocal: bit-identical to the reference's division), otherwise like four rows.
    // FEAT bit 13: ... and of MIXED lengths one to four rows (a 6-hourly series with missing steps, a 12-hourly one joined to a
    //              6-hourly one): every group owns four row registers and fills as many as it is long; its length is a scalar read from
    //              the group table, the loads and the statistics' tail rows sit under scalar branches on it, the mean is div_by's
    //              correctly rounded s / n (n = 2, 4: exact anyway).  Otherwise the four-row form.
    constexpr bool RAG = (FEAT & 8192) != 0;
    constexpr int GL = ((FEAT & 1024) || RAG) ? 4 : ((FEAT & 4096) ? 3 : 2);
    static_assert(!(FEAT & (1024 | 4096 | 8192)) || (PAIR && LEAN && !LEAN_SINE), "three- / four-row / mixed groups: a lean short-group form");
    static_assert(((FEAT & 1024) != 0) + ((FEAT & 4096) != 0) + ((FEAT & 8192) != 0) <= 1, "one group length rule per variant");
    static_assert(!PAIR || (PIPE == 0 && (STAT == 2 || (STAT == 1 && LEAN)) && NTHR == 0 && DEPTH % GL == 0),
                  "short-group mode: direct loads, sum (+ min + max), no threshold slots");
    static_assert(!LEAN || PAIR, "the lean group end is a short-group form");
    // FEAT bit 11: region-fused period ends compiled in (FusedArgs::rf_w).  A twin of the plain variant: the staging code at the
    // period end raises the register count by ~7 (one wave per SIMD less on the float32 two-cell forms), so only plans that take
    // the route run it; with rf_w == null it behaves like its twin.
    constexpr bool RF = (FEAT & 2048) != 0;
    static_assert(!RF || !(SL || HB), "region-fused period ends: two-level plans only");
    const int64_t C = a.C;
    const int K = a.K;
    const int lane = threadIdx.x & 63;
    // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  With xcd_remap the
    // tiles of one XCD form a contiguous run of cells; it changes which XCD reads what, never
    // what is computed (bijective for any tile count).
    int64_t tile = blockIdx.x;
    if (a.xcd_remap) {
        const int nt = a.n_tiles, q = nt / 8, r = nt % 8, x = (int)(blockIdx.x % 8), i = (int)(blockIdx.x / 8);
        tile = (x < r ? (int64_t)x * (q + 1) : (int64_t)r * (q + 1) + (int64_t)(x - r) * q) + i;
    }
    const int64_t c0 = (tile * blockDim.x + threadIdx.x) * VEC;
    const bool active = c0 < C;
    const int64_t c_ld = active ? c0 : (C - VEC);   // clamped: inactive lanes re-read valid cells
    int64_t k_lo, k_hi;
    int g_lo, g_hi, slot;
    {
        const int64_t* w = (const int64_t*)&a.chunks[blockIdx.y];
        k_lo = ld_uniform(w);
        k_hi = ld_uniform(w + 1);
        const int64_t g = ld_uniform(w + 2), sb = ld_uniform(w + 3);
        g_lo = (int32_t)(g & 0xffffffffLL); g_hi = (int32_t)(g >> 32);
        slot = (int32_t)(sb & 0xffffffffLL);
    }
    const int rows = (int)(k_hi - k_lo);            // chunk-relative 32-bit loop control: scalar ALU only
    const TIn* __restrict__ cube = (const TIn*)a.cube + k_lo * C + c_ld;

    // ---- per-cell state, all in registers (compile-time indexed) ----
    double s[VEC], mn[VEC], mx[VEC];
    int cnt[VEC];
    unsigned long long nanmask[VEC];                // lane masks in SGPR pairs: OR-ed on the scalar ALU
    bool pnan[VEC];                                 // pair mode: this lane's pair holds a NaN
    TIn plo[VEC], phi[VEC];                         // pair mode: the pair's min / max in the input precision
    unsigned long long nanacc[VEC];                 // lean group end: lanes that met a NaN pair since the last emitted slot
    double acc[(NTHR > 0 && !TKI) ? NTHR : 1][VEC];
    int cthr[(NTHR > 0 && TKI && !HB) ? NTHR : 1][VEC];
    double os[SL ? 1 : KMAX][VEC];

#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        s[i] = 0.0; mn[i] = inf64(); mx[i] = -inf64(); cnt[i] = 0; nanmask[i] = 0ull; nanacc[i] = 0ull;
#pragma unroll
        for (int j = 0; j < NTHR; ++j) { if (HB) {} else if (TKI) cthr[j][i] = 0; else acc[j][i] = 0.0; }
    }
    // LDS-histogram state.  With E[0] = E[hb_n + 2] = NaN (sentinels: every compare fails) and
    // E[k + 1] = edge k, bin g of the guarded partition (g = 0 and g = hb_n + 1 are the guard bins) lies
    // between E[g] and E[g + 1]:
    //   etab_a[g] = {E[g].dn, E[g+1].up}   what the common path needs: v <= dn -> below, v >= up -> above
    //   etab_b[g] = {E[g].up, E[g+1].dn}   only for lanes that leave the guessed bin: is v ON the edge?
    // then the counters [(hb_n + 2) * VEC][blockDim].
    struct EdgeT { TIn lo, hi; };
    extern __shared__ __attribute__((aligned(32))) unsigned char dynlds[];
    EdgeT* etab_a = (EdgeT*)dynlds;
    EdgeT* etab_b = (EdgeT*)(dynlds + HB_TABLE_BYTES / 2);
    int* hcnt = (int*)(dynlds + HB_TABLE_BYTES);
    const int bd = blockDim.x, tid = threadIdx.x;
    // sine_dd plans: every workgroup copies the acos table (sine_theta) into LDS, behind the LDS-DMA ring if there is one
    static_assert(!(HB && (FEAT & 1)), "histogram variants carry no sine_dd code");
    // (pair-mode variants: the G table of sine_pair_g; the others: the acos table of sine_theta — the host hands over the one
    // the variant reads)
    sine_tab_t sine_tab = nullptr;
    sine_p2_t sine_p2 = nullptr;
    if constexpr ((FEAT & 1) != 0) {
        unsigned char* base = dynlds + (PIPE == 1 ? (size_t)(bd >> 6) * DEPTH * 1024 : (size_t)0);
        if (a.sine_tab != nullptr) {            // uniform: the host sets it iff a column is sine_dd (and then sizes the LDS for it)
            typedef double d2 __attribute__((ext_vector_type(2)));
            constexpr int bytes = ((FEAT & 128) != 0 && (FEAT & (1024 | 4096)) == 0) ? SINE_P2_BYTES : SINE_TAB_BYTES;
            // (the sine-only lean form works in DOUBLED units — clamp, arcs and the column's sum — and halves once per period end:
            // its copy of the table is 2 H, every step an exact scaling of the undoubled one)
            for (int e = tid; e < bytes / 16; e += bd) {
                d2 t = ((const d2*)a.sine_tab)[e];
                if constexpr (LEAN_SINE) t *= 2.0;
                ((d2*)base)[e] = t;
            }
            __syncthreads();
        }
        sine_tab = (sine_tab_t)(lds_ptr_t)base;
        sine_p2 = (sine_p2_t)(lds_ptr_t)base;
    }
    const int hb_bins = a.hb_n + 2;
    TIn hb_c1 = 0, hb_c0 = 0, hb_top = 0;
    TIn ha_w = 0, ha_c0b = 0, ha_e0 = 0, ha_gl = 0, ha_gh = 0;
    int hb_sh = 0, hb_lane[VEC] = {0};
    if constexpr (HB) {
        if (tid < hb_bins) {
            auto edge = [&](int k, bool want_up) -> TIn {        // E[k].up or E[k].dn
                if (k == 0 || k == a.hb_n + 2) return (TIn)nan64();
                if constexpr (sizeof(TIn) == 4) return want_up ? a.hb_up[k - 1] : a.hb_dn[k - 1];
                else return a.hb_edge[k - 1];
            };
            EdgeT ea, eb;
            ea.lo = edge(tid, false); ea.hi = edge(tid + 1, true);
            eb.lo = edge(tid, true);  eb.hi = edge(tid + 1, false);
            etab_a[tid] = ea; etab_b[tid] = eb;
        }
        for (int b = 0; b < hb_bins * VEC; ++b) hcnt[b * bd + tid] = 0;
        __syncthreads();
        if constexpr (sizeof(TIn) == 4) { hb_c1 = a.hb_c1f; hb_c0 = a.hb_c0f; }
        else { hb_c1 = a.hb_c1; hb_c0 = a.hb_c0; }
        hb_top = (TIn)(a.hb_n + 1);
        if constexpr (HA) {
            if constexpr (sizeof(TIn) == 4) { ha_w = a.hb_wf; ha_e0 = a.hb_lo0f + a.hb_wf; ha_gl = a.hb_glf; ha_gh = a.hb_ghf; ha_c0b = a.hb_c0bf; }
            else { ha_w = a.hb_w; ha_e0 = a.hb_lo0 + a.hb_w; ha_gl = a.hb_gl; ha_gh = a.hb_gh; ha_c0b = a.hb_c0b; }   // (lo0 + w: exact, host-checked)
            // VOP3 takes one scalar operand: keep the second operand of the clamp and the fma addends in VGPRs
            asm volatile("" : "+v"(ha_gh), "+v"(ha_c0b), "+v"(ha_e0));
        }
        asm volatile("" : "+v"(hb_c0));      // keep the addend in a VGPR: v_fma takes one scalar operand only
        hb_sh = a.hb_shift + 2 + (VEC == 4 ? 2 : (VEC == 2 ? 1 : 0));     // byte stride between bins = VEC * blockDim * 4
#pragma unroll
        for (int i = 0; i < VEC; ++i) hb_lane[i] = ((i << a.hb_shift) + tid) * 4;
    }
    auto reset_outer = [&]() {
        if (SL) return;
        if constexpr ((FEAT & 256) != 0) {          // lean forms: every outer reducer is sum | mean — no per-column identity, no j < K masks
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
#pragma unroll
                for (int i = 0; i < VEC; ++i) os[j][i] = 0.0;
            return;
        }
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (j < K) {
                const int o = a.cols[j].outer;
                const double init = (o == OUT_MIN) ? inf64() : ((o == OUT_MAX) ? -inf64() : 0.0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) os[j][i] = init;
            }
        }
    };
    reset_outer();

    // ---- region-fused period ends (FusedArgs::rf_w): what a wave needs at every period end, read once ----
    // A wave's tile is its 64 * VEC consecutive cells.  For entry e (the cell's first / second region) a RUN is a maximal stretch of
    // cells of the tile with the same region; the host numbered the runs of a tile in cell order and worked out, per lane, how the
    // segmented scan of the period end proceeds (afhip_api.hip: rf_table — the tables depend on the weights table and VEC only):
    //   weights     of cell i's entries e = 0, 1 (0: none)
    //   bits        per lane, 16 bits per entry e:  bits 0-5   step s of the segmented scan adds the value of lane - 2^s
    //                                                bits 6+i   cell i starts a stretch (region changes, no region, or the tile starts)
    //                                                bits 8+i   cell i ends a run of a region: its sum is stored
    //   rid         byte 2 e + i: index of cell i's run among the runs of (tile, e)
    //   rf_first[e] first run of (tile, e) in rf_out
    //   rf_need[e]  (uniform) bit s: some lane of the wave adds in step s — steps no run of this tile is long enough for are skipped;
    //               bit 6: some run ends at a lane's first cell (two cells per lane); bit 7: the tile has runs of entry e at all
    static_assert(!RF || VEC <= 2, "region-fused period ends: one or two cells per lane");
#ifndef AFHIP_RF_CB
#define AFHIP_RF_CB 2
#endif
    // columns per block of the period end's scan (registers against overlap).  The general two-cell forms of six columns take ONE (AFHIP_RF_SLIM):
    // they are held by their bytes in flight, and with the weights re-read per block and the store addresses formed at the stores the twin fits six waves per SIMD
#ifndef AFHIP_RF_SLIM
#define AFHIP_RF_SLIM 1
#endif
    constexpr bool RF_SLIM = AFHIP_RF_SLIM && VEC == 2 && KMAX == 6 && !(FEAT & 128) && sizeof(TIn) == 4;
    constexpr int RF_CB = RF_SLIM ? 1 : AFHIP_RF_CB;
    // The lane's weights and words are PARKED in LDS (a wave-private block behind the variant's other LDS: RF_LANE_BYTES per lane) and
    // read back at every period end: held in registers they cost the twins ten VGPRs for the whole kernel — 103 against the plain
    // variant's 74 on the float32 configs[1] plan, four waves per SIMD instead of six, for a kernel that is bound by bytes in flight.
    constexpr int RF_LANE_BYTES = VEC * 16 + 16;      // (a multiple of 16: the block is read and written in 16-byte pieces)
    typedef __attribute__((address_space(3))) unsigned char* rf_lds_t;
    rf_lds_t rf_park = nullptr;
    int rf_first[2] = {0, 0}, rf_need[2] = {0, 0};
    if constexpr (RF) {
        if (a.rf_w != nullptr) {
            const int64_t lane0 = (c0 - (int64_t)lane * VEC) / VEC;                      // the wave's first lane slot (uniform)
            const int64_t wt = (int64_t)__builtin_amdgcn_readfirstlane((int)(lane0 / 64));
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            typedef double d2 __attribute__((ext_vector_type(2)));
            const u2 lw = *(const u2*)(a.rf_lane + (wt * 64 + lane) * 2);                 // (the table is padded to whole tiles)
            rf_park = (rf_lds_t)(lds_ptr_t)(dynlds + a.rf_lds_off) + (threadIdx.x * RF_LANE_BYTES);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                d2 w = d2{0.0, 0.0};
                if (active) w = *(const d2*)(a.rf_w + (c0 + i) * 2);
                *(__attribute__((address_space(3))) d2*)(rf_park + i * 16) = w;
            }
            *(__attribute__((address_space(3))) u2*)(rf_park + VEC * 16) = lw;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                rf_first[e] = ld_uniform(a.rf_tile + (wt * 2 + e) * 2);
                rf_need[e] = ld_uniform(a.rf_tile + (wt * 2 + e) * 2 + 1);
            }
        }
    }

    // ---- the hot per-element update ----
    // LDS-histogram update of one value, in two halves so that a batch of rows can issue all its
    // table reads before the first compare needs one (see the PIPE 0 loop).
    auto hb_guess = [&](TIn vr, int& b, EdgeT& ea) {
        // guess in the input precision, clamped into the guarded range while still a float:
        // fmax(NaN, 0) = 0, so a NaN drops into the lower guard bin without a test of its own;
        // the clamped value is >= 0, so the truncating conversion is the floor
        TIn t;
        if constexpr (sizeof(TIn) == 4) t = __builtin_amdgcn_fmed3f(__fmaf_rn(vr, hb_c1, hb_c0), 0.0f, hb_top);
        else t = fmin(fmax(__fma_rn((double)vr, hb_c1, hb_c0), 0.0), hb_top);
        b = (int)t;
        ea = etab_a[b];
    };
    auto hb_count = [&](TIn vr, int i, int b, const EdgeT& ea) {
        const bool up = vr >= ea.hi, dn = vr <= ea.lo;               // the value belongs above / below the guess
        bool count = true;
        if (up || dn) {                                               // rare: guess off by one, or v on an edge
            const EdgeT eb = etab_b[b];
            count = up ? !(vr <= eb.hi) : !(vr >= eb.lo);             // strict bins: an edge value is in none
            b += up ? 1 : -1;
        }
        // the counter is private to this lane: a relaxed LDS atomic is one ds_add_u32 (no return value)
        if (count)
            __hip_atomic_fetch_add((int*)((char*)hcnt + (b << hb_sh) + hb_lane[i]), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    // arithmetic-edge form of hb_guess + hb_count: no table read, and (round 3) a ONE-SIDED guess.  With the guess constant biased
    // down by a host-chosen delta, tf = floor(vc * c1 + c0b) is the value's bin or the one below it, and exactly the one below for a
    // value ON an edge (FusedArgs::hb_c0b: the host checks every edge with this very fma; fma and floor are monotone, so what holds
    // at E[k] and E[k + 1] holds between them).  Only the UPPER edge of the guessed bin is then needed, hi = E[tf + 1] (one exact
    // fma):  vc > hi -> bin tf + 1;  vc == hi -> on an edge, in no bin (strict inequalities, nb_kernels.py:190-196);  else bin tf.
    // 9 VALU per element where the two-sided form (both edges, two repairs, two equality tests) took 14: the kernel had become
    // issue-bound beside its stream (16 VALU per 4-byte element with the address arithmetic: the vector ALUs ~70 % busy,
    // profiles/r03_c4_bound_pmc.txt).
    auto ha_update = [&](TIn vr, int i) {
        TIn vc;                                                     // clamped into the guarded range; NaN -> lower guard bin
        if constexpr (sizeof(TIn) == 4) vc = __builtin_amdgcn_fmed3f(vr, ha_gl, ha_gh);
        else vc = fmin(fmax(vr, ha_gl), ha_gh);
        TIn tf, hi;
        if constexpr (sizeof(TIn) == 4) {
            tf = __builtin_floorf(__fmaf_rn(vc, hb_c1, ha_c0b));
            hi = __fmaf_rn(tf, ha_w, ha_e0);                        // E[tf + 1], exactly
        } else {
            tf = __builtin_floor(__fma_rn(vc, hb_c1, ha_c0b));
            hi = __fma_rn(tf, ha_w, ha_e0);
        }
        int b = (int)tf;
        b += (vc > hi) ? 1 : 0;                                     // (a carry-in add)
        if (vc != hi) {
            int* p = (int*)((char*)hcnt + (b << hb_sh) + hb_lane[i]);
#ifdef HA_PLAIN_RMW
            *p = *p + 1;                                              // the counter is private to this lane
#else
            __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
        }
    };
    auto consume = [&](const RawVec<TIn, VEC>& rv, bool hb_inline = true) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const TIn vr = rv.v[i];
            const double v = (double)vr;
            const bool isn = vr != vr;
            // integer-bin plans without an inner statistic never look at the NaN mask (a NaN is simply in no bin)
            if constexpr (!(TKI && STAT == 0)) nanmask[i] |= __builtin_amdgcn_ballot_w64(isn);
            if (STAT == 1) {
                s[i] += v;                          // a NaN poisons s; the group is NaN anyway
            } else if (STAT == 2) {
                s[i] += v;
                mn[i] = __builtin_fmin(mn[i], v);   // minNum/maxNum skip a NaN like the reference's
                mx[i] = __builtin_fmax(mx[i], v);   // "v < mn" / "v > mx" updates: one VALU op each
            } else if (STAT == 3) {
                s[i] += isn ? 0.0 : v;
                cnt[i] += isn ? 0 : 1;
                mn[i] = __builtin_fmin(mn[i], v);
                mx[i] = __builtin_fmax(mx[i], v);
            }
            if constexpr (HB) {
                if constexpr (HA) { if (hb_inline) ha_update(vr, i); }
                else if (hb_inline) { int b; EdgeT ea; hb_guess(vr, b, ea); hb_count(vr, i, b, ea); }
            }
#pragma unroll
            for (int j = 0; j < (HB ? 0 : NTHR); ++j) {
                if constexpr (TKI) {
                    bool m;                                               // strict, NaN -> false
                    if constexpr (sizeof(TIn) == 4) m = (vr > a.thr[j].t0f) && (vr < a.thr[j].t1f);
                    else m = (v > a.thr[j].t0) && (v < a.thr[j].t1);
                    cthr[j][i] += m ? 1 : 0;
                } else {
                    const double w = __fma_rn(a.thr[j].A, v, a.thr[j].B);
                    if constexpr (sizeof(TIn) == 4) add_if_between<float>(acc[j][i], w, vr, a.thr[j].t0f, a.thr[j].t1f);
                    else add_if_between<double>(acc[j][i], w, v, a.thr[j].t0, a.thr[j].t1);
                }
            }
        }
    };

    // ---- region-fused period end (FusedArgs::rf_w): the period's K values of this lane's cells -> per-run weighted sums ----
    // val[j][i]: column j of cell i (NaN = missing).  The lanes multiply their cells' values (where(valid, x, 0), and the valid flag as
    // column K) by the cells' weights; a segmented inclusive scan over the lanes — six shuffle steps whose add / skip pattern was
    // fixed at kernel start, of which the wave runs only those some run of its tile needs — carries a run's sum to its last cell,
    // and that lane stores it.  The order of the adds is a fixed tree over the cells of the tile: it depends on where the table's
    // runs lie, never on the data or the launch shape.  Products are rounded before the adds (spatial.py:183-185).
    // (Round 3 staged the values in a wave-private LDS block and let one lane per (run, column) walk its run: a serial chain of LDS
    // round trips per period end, and the weights re-read from memory at every one — 3.7 against 3.3 ms on the daily configs[1]
    // panel, 27.5 against 17.7 on a daily sine_dd panel of (tmin, tmax) pairs; profiles/r04_region_fused_scan.txt.)
    auto rf_emit = [&](double (&val)[KMAX][VEC], int at_slot) {
        const int K1 = K + 1;
        // shared validity, applied in place (the caller resets / drops its period values right after): x = where(valid, x, 0)
        bool ok[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            bool valid = active;                                     // (lanes beyond the grid carry no weight and no run)
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
                if (j < K) valid = valid && (val[j][i] == val[j][i]);
            ok[i] = valid;
            if (!valid) {
                KEEP_BRANCH();
#pragma unroll
                for (int j = 0; j < KMAX; ++j) val[j][i] = 0.0;
            }
            if (a.rf_x != nullptr) {                                 // (uniform) the table has cells in three or more regions
                const int xi = active ? a.rf_x[c0 + i] : -1;
                if (xi >= 0) {                                       // rare lanes: the cell's values for its extra entries
                    double* ex = a.rf_ex + ((int64_t)at_slot * a.rf_nx + xi) * K1;
#pragma unroll
                    for (int j = 0; j < KMAX; ++j)
                        if (j < K) ex[j] = val[j][i];
                    ex[K] = ok[i] ? 1.0 : 0.0;
                }
            }
        }
        // (the lanes' bit fields are made opaque here: tested where they are used, the compiler would otherwise hoist all the lane
        // masks they decode to out of the time loop — two dozen scalar register pairs that spilled into v_readlane traffic)
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 lw = *(const __attribute__((address_space(3))) u2*)(rf_park + VEC * 16);
        uint32_t rb = lw.x, rr = lw.y;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("; rf_emit: begin" : "+v"(rb), "+v"(rr));
#endif
        // Per entry: the lane conditions and store addresses once, then the columns in blocks of CB.  The column list is fixed at
        // compile time — the validity weight first, then val[0 .. KMAX) — so nothing inside the scan branches on K: columns beyond K
        // carry whatever their registers hold through the shuffles and only their stores are skipped (the weight is stored at index K
        // of the record).  Per block: products, the lane's own stretch, the scan steps some run of the tile needs (a shuffle = two
        // ds_bpermute on an address formed once per step; the adds run under the lanes' take / start / end bits as branches on lane
        // conditions, i.e. EXEC masks, not selects), the sums of the runs that end in this lane.
        constexpr int CB = RF_CB, NCOL = KMAX + 1;
        auto shfl64 = [&](double x, int addr) -> double {
            const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
            const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
            return __hiloint2double(hi, lo);
        };
#if AFHIP_RF_DPP
        // the scan's data movement as DPP moves (no LDS round trip): steps 0 - 3 shift by 1, 2, 4, 8 lanes inside each row of 16
        // (row_shr), step 4 hands every row's last lane to the next row (row_bcast:15 -> rows 1 and 3), step 5 lane 31 to rows 2 and 3
        // (row_bcast:31); `carried` is the whole wave shifted by one lane (wave_shr:1).  Lanes without a source keep 0 — they never add.
        auto dpp64 = [&](double x, int st) -> double {
            const int xl = __double2loint(x), xh = __double2hiint(x);
            int lo, hi;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "=v"(lo));                             // (lanes a move does not write keep what the register held: no zeroing —
            asm volatile("" : "=v"(hi));                             // they are the lanes that do not add at this step)
#else
            lo = hi = 0;
#endif
            switch (st) {
                case 0: lo = __builtin_amdgcn_update_dpp(lo, xl, 0x111, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, xh, 0x111, 0xf, 0xf, false); break;
                case 1: lo = __builtin_amdgcn_update_dpp(lo, xl, 0x112, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, xh, 0x112, 0xf, 0xf, false); break;
                case 2: lo = __builtin_amdgcn_update_dpp(lo, xl, 0x114, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, xh, 0x114, 0xf, 0xf, false); break;
                case 3: lo = __builtin_amdgcn_update_dpp(lo, xl, 0x118, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, xh, 0x118, 0xf, 0xf, false); break;
                case 4: lo = __builtin_amdgcn_update_dpp(lo, xl, 0x142, 0xa, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, xh, 0x142, 0xa, 0xf, false); break;
                case 5: lo = __builtin_amdgcn_update_dpp(lo, xl, 0x143, 0xc, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, xh, 0x143, 0xc, 0xf, false); break;
                default: lo = __builtin_amdgcn_update_dpp(lo, xl, 0x138, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, xh, 0x138, 0xf, 0xf, false); break;
            }
            return __hiloint2double(hi, lo);
        };
#endif
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int need = rf_need[e];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+s"(need));                           // (the same for the wave's step mask: s_bitcmp here, not a spilled mask per step)
#endif
            if (!(need & 128)) continue;                             // (uniform) no cell of this tile has an entry e
            const uint32_t bits = rb >> (16 * e), rid = rr >> (16 * e);
            double rfw0[VEC];                                        // the cells' weights for this entry, from the lane's parking block
            if constexpr (!RF_SLIM) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) rfw0[i] = *(const __attribute__((address_space(3))) double*)(rf_park + i * 16 + e * 8);
            }
            double* out0 = a.rf_out + (int64_t)at_slot * a.rf_slot_stride + (int64_t)rf_first[e] * a.rf_run_stride;
            double* dst0_ = nullptr; double* dst1_ = nullptr;
            if constexpr (!RF_SLIM) {
                dst0_ = out0 + (int64_t)(rid & 0xffu) * a.rf_run_stride;              // where this lane's cells store, if they end a run
                dst1_ = out0 + (int64_t)((rid >> 8) & 0xffu) * a.rf_run_stride;
            }
#define dst0 (RF_SLIM ? out0 + (int64_t)(rid & 0xffu) * a.rf_run_stride : dst0_)
#define dst1 (RF_SLIM ? out0 + (int64_t)((rid >> 8) & 0xffu) * a.rf_run_stride : dst1_)
            const bool joins = !((bits >> 7) & 1u), fresh = ((bits >> 6) & 1u) != 0u, end0 = ((bits >> 8) & 1u) != 0u, end1 = ((bits >> 9) & 1u) != 0u;
#pragma unroll
            for (int q0 = 0; q0 < NCOL; q0 += CB) {                  // q: position in the column list (0 = the weight, q >= 1: val[q - 1])
                if (q0 > K) continue;                                // (uniform: the block's first value column is val[q0 - 1])
                double v[CB], p0[CB];
                double rfw[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) rfw[i] = RF_SLIM ? *(const __attribute__((address_space(3))) double*)(rf_park + i * 16 + e * 8) : rfw0[i];
#pragma unroll
                for (int qq = 0; qq < CB; ++qq) {
                    const int q = q0 + qq;
                    v[qq] = 0.0; p0[qq] = 0.0;
                    if (q < NCOL) {
                        // (the validity weight's product w * 1 / w * 0 is a select)
                        p0[qq] = q == 0 ? (ok[0] ? rfw[0] : 0.0) : __dmul_rn(rfw[0], val[q == 0 ? 0 : q - 1][0]);
                        v[qq] = VEC == 2 ? (q == 0 ? (ok[VEC - 1] ? rfw[VEC - 1] : 0.0) : __dmul_rn(rfw[VEC - 1], val[q == 0 ? 0 : q - 1][VEC - 1])) : p0[qq];
                    }
                }
                if constexpr (VEC == 2) {
                    if (joins) {                                     // (lanes whose second cell continues the first cell's stretch)
                        KEEP_BRANCH();
#pragma unroll
                        for (int qq = 0; qq < CB; ++qq)
                            if (q0 + qq < NCOL) v[qq] = __dadd_rn(p0[qq], v[qq]);
                    }
                }
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    if (!((need >> st) & 1)) continue;               // (uniform)
                    double pv[CB];
#if AFHIP_RF_DPP
#pragma unroll
                    for (int qq = 0; qq < CB; ++qq) pv[qq] = q0 + qq < NCOL ? dpp64(v[qq], st) : 0.0;
#else
                    const int addr = ((lane - (1 << st)) & 63) << 2;
#pragma unroll
                    for (int qq = 0; qq < CB; ++qq) pv[qq] = q0 + qq < NCOL ? shfl64(v[qq], addr) : 0.0;
#endif
                    if ((bits >> st) & 1u) {
                        KEEP_BRANCH();
#pragma unroll
                        for (int qq = 0; qq < CB; ++qq)
                            if (q0 + qq < NCOL) v[qq] = __dadd_rn(v[qq], pv[qq]);
                    }
                }
                // record index of list position q: the weight at K, val[q - 1] at q - 1 (stored when q - 1 < K)
                if constexpr (VEC == 2) {
                    if (need & 64) {                                 // (uniform) a run ends at some lane's FIRST cell: what the previous lanes
                        const int addr = ((lane - 1) & 63) << 2;     // carried (unless the cell starts a stretch) + the cell
                        double carried[CB];
#pragma unroll
#if AFHIP_RF_DPP
                        for (int qq = 0; qq < CB; ++qq) carried[qq] = q0 + qq < NCOL ? dpp64(v[qq], 6) : 0.0;
#else
                        for (int qq = 0; qq < CB; ++qq) carried[qq] = q0 + qq < NCOL ? shfl64(v[qq], addr) : 0.0;
#endif
                        if (end0) {
                            KEEP_BRANCH();
                            if (!fresh) {
                                KEEP_BRANCH();
#pragma unroll
                                for (int qq = 0; qq < CB; ++qq)
                                    if (q0 + qq < NCOL) p0[qq] = __dadd_rn(carried[qq], p0[qq]);
                            }
#pragma unroll
                            for (int qq = 0; qq < CB; ++qq) {
                                const int q = q0 + qq;
                                if (q == 0) dst0[K] = p0[qq];
                                else if (q < NCOL && q - 1 < K) dst0[q - 1] = p0[qq];
                            }
                        }
                    }
                    if (end1) {
                        KEEP_BRANCH();
#pragma unroll
                        for (int qq = 0; qq < CB; ++qq) {
                            const int q = q0 + qq;
                            if (q == 0) dst1[K] = v[qq];
                            else if (q < NCOL && q - 1 < K) dst1[q - 1] = v[qq];
                        }
                    }
                } else {
                    if (end0) {
                        KEEP_BRANCH();
#pragma unroll
                        for (int qq = 0; qq < CB; ++qq) {
                            const int q = q0 + qq;
                            if (q == 0) dst0[K] = v[qq];
                            else if (q < NCOL && q - 1 < K) dst0[q - 1] = v[qq];
                        }
                    }
                }
            }
        }
#undef dst0
#undef dst1
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("; rf_emit: end");
#endif
    };

    // ---- end of an inner group: column values, transforms, outer accumulation ----
    auto group_end = [&](bool emit_slot, int nsteps, double inv_n, int zoff, int gidx) {
        const bool empty = nsteps == 0;
        const double dn = (double)nsteps;
        bool hasnan[VEC];
        double mean[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if constexpr (PAIR) hasnan[i] = pnan[i];
            else hasnan[i] = ((nanmask[i] >> lane) & 1ull) != 0ull;
            if constexpr (PAIR && GL == 3) mean[i] = div_by(s[i], 3.0, 1.0 / 3.0);      // == s / 3 bit for bit (1.0 / 3.0: the correctly rounded reciprocal)
            else if constexpr (PAIR && RAG) mean[i] = div_by(s[i], dn, inv_n);          // the group's own length (1 .. 4)
            else if constexpr (PAIR) mean[i] = s[i] * (1.0 / GL);      // == s / 2 (s / 4) bit for bit
            else mean[i] = (STAT >= 1) ? div_by(s[i], dn, inv_n) : 0.0;      // == s / dn bit for bit (inv_n = RN(1/n))
        }
        // single-sine degree days of one column (nb_kernels.py:218-251).  The reciprocal of the window's range and the arcs are
        // only evaluated where a threshold lies strictly inside (tmin, tmax): a branch on a lane condition skips the whole wave
        // when no lane needs it (coherent real data: most wave-days), and the other two cases are one subtraction.
        auto sine_column = [&](const ColOp& co, double (&x)[VEC]) {
            const double INV_PI = 0.31830988618379067154;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const double tavg = mean[i];
                bool in0 = false, in1 = false;
                if constexpr (PAIR && GL == 2 && LEAN) {
                    // (the lean pair form tests its windows on 2 thr - s, below)
                } else if constexpr (PAIR && GL == 2 && sizeof(TIn) == 4) {       // float data: exact float compares against the rounded thresholds
                    in0 = (plo[i] < co.s0up) && (phi[i] > co.s0dn);
                    in1 = (plo[i] < co.s1up) && (phi[i] > co.s1dn);
                } else {
                    in0 = (mn[i] < co.s0) && (co.s0 < mx[i]);
                    in1 = (mn[i] < co.s1) && (co.s1 < mx[i]);
                }
                double xv;
                if constexpr (PAIR && GL == 2 && LEAN) {
                    // the lean form in DOUBLED units: D = 2 (thr - tavg) = 2 thr - s with one rounding (= 2 d bit for bit), rng = 2 alpha
                    // exact, so  tmin < thr < tmax  <=>  |D| < rng  is ONE compare per threshold instead of two (were RN(2 thr - s)
                    // to land on rng from inside, half an ulp away, the arc it skips is exactly 0: u2 = RN(rng - |D|) = 0), u2 is one
                    // add, and the column's two max() terms are one clamp of width 2 (s1 - s0), halved exactly:
                    // max(t - s0, 0) - max(t - s1, 0) = clamp(t - s0, 0, s1 - s0) — the reference's (t - s0) - (t - s1) for t >= s1 is
                    // s1 - s0 up to its own two roundings (1e-16 relative; sine_dd's contract is 1e-10)
                    // (rd: the sine-only form keeps the pair unordered — its min and max are never formed — so the range is |rd|, an abs
                    // modifier on every instruction that reads it)
                    const double rd = mx[i] - mn[i];
                    const double D0 = __fma_rn(s[i], -1.0, co.s0x2), D1 = __fma_rn(s[i], -1.0, co.s1x2);
                    const bool i0 = fabs(D0) < fabs(rd), i1 = fabs(D1) < fabs(rd);
                    double c2;
                    if (co.skind == 0) { KEEP_BRANCH(); c2 = min_vs(max0_neg(D0), co.swidth2); }
                    else c2 = min_vs(max0(D1), co.swidth2);
                    // (folding the halving and the day's add into the column's sum into one fma: built, level; r03_session2_experiments.txt)
                    if constexpr (LEAN_SINE) xv = c2;       // doubled units all the way: the table in LDS is 2 H, the emit halves
                    else xv = c2 * 0.5;
                    if (i0 || i1) {
                        // cooling: + part(s0) - part(s1); heating: the reverse; the sign rides on u2 = +-(rng - |D|)
                        double w, p, su;
                        if (i0) {
                            if (co.skind == 0) { KEEP_BRANCH(); su = fabs(rd) - fabs(D0); } else su = fabs(D0) - fabs(rd);
                            sine_pair_g(su, rd, sine_p2, w, p);
                            xv = __fma_rn(w, p, xv);
                        }
                        if (i1) {
                            if (co.skind == 0) { KEEP_BRANCH(); su = fabs(rd) - fabs(D1); } else su = fabs(D1) - fabs(rd);
                            sine_pair_g(su, rd, sine_p2, w, p);
                            xv = __fma_rn(-w, p, xv);
                        }
                    }
                } else if constexpr (PAIR && GL == 2) {
                    // tavg is the mid-range: part = max(+-(tavg - thr), 0) + [inside] alpha F(|thr - tavg| / alpha)  (sine_pair_g)
                    // thr - tavg = thr - s / 2 (s / 2 is exact: one rounding either way)
                    const double d0 = __fma_rn(s[i], -0.5, co.s0), d1 = __fma_rn(s[i], -0.5, co.s1);
                    if (co.skind == 0) { KEEP_BRANCH(); xv = max0_neg(d0) - max0_neg(d1); }
                    else xv = max0(d1) - max0(d0);
                    if (in0 || in1) {
                        const double rng = mx[i] - mn[i];               // = 2 alpha, exact
                        // cooling: + part(s0) - part(s1); heating: the reverse.  (ONE arc site for both thresholds — a lane's window
                        // rarely holds both, but some lanes of a wave hold s0 while others hold s1 on spring / autumn days — was built
                        // and measured: only 7 % fewer arcs on the ERA5-like field, and the operand selects cost more: 23.1 -> 24.9
                        // VALU per cell-step, same time; profiles/r03_kbench_c5_table_arc.txt)
                        // (the sign rides on u2 = rng - 2 |d|: a scalar branch around one instruction instead of a multiplication)
                        double w, p, su;
                        if (in0) {
                            if (co.skind == 0) { KEEP_BRANCH(); su = __fma_rn(fabs(d0), -2.0, rng); } else su = __fma_rn(fabs(d0), 2.0, -rng);
                            sine_pair_g(su, rng, sine_p2, w, p);
                            xv = __fma_rn(w, p, xv);
                        }
                        if (in1) {
                            if (co.skind == 0) { KEEP_BRANCH(); su = __fma_rn(fabs(d1), -2.0, rng); } else su = __fma_rn(fabs(d1), 2.0, -rng);
                            sine_pair_g(su, rng, sine_p2, w, p);
                            xv = __fma_rn(-w, p, xv);
                        }
                    }
                } else {
                    const double rng = mx[i] - mn[i], alpha = rng * 0.5;
                    double inv_rng = 0.0;
                    if (in0 || in1) inv_rng = rcp_newton1(rng);
                    if (co.skind == 0) {
                        KEEP_BRANCH();
                        xv = sine_cool(co.s0, co.s0x2, in0, mn[i], mx[i], tavg, alpha, inv_rng, sine_tab)
                           - sine_cool(co.s1, co.s1x2, in1, mn[i], mx[i], tavg, alpha, inv_rng, sine_tab);
                    } else {
                        xv = -sine_heat(co.s0, in0, mn[i], mx[i], tavg, alpha, 2.0 * inv_rng, sine_tab)
                           + sine_heat(co.s1, in1, mn[i], mx[i], tavg, alpha, 2.0 * inv_rng, sine_tab);
                    }
                }
                if constexpr (LEAN) {
                    x[i] = xv;                                  // the NaN pairs are kept in nanacc (below)
                } else {
                    // a NaN window: any NaN will do — only the high word is replaced (one v_cndmask instead of two)
                    const int hi = (hasnan[i] || empty) ? 0x7ff80000 : __double2hiint(xv);
                    x[i] = __hiloint2double(hi, __double2loint(xv));
                }
            }
        };
        if constexpr (LEAN) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) nanacc[i] |= __builtin_amdgcn_ballot_w64(pnan[i]);
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                if (j < K) {
                    double x[VEC];
                    if constexpr (LEAN_SINE) {
                        sine_column(a.cols[j], x);              // a.cols[j]: loop-invariant kernel argument, scalar registers
                    } else {
                        // (the empty asm makes the word opaque at this point: the six words stay in scalar registers through the time
                        // loop while their decoded fields — a multiplier, masks and a trip count per column — are NOT hoisted out of
                        // it, where they overflowed the scalar register file and came back through v_readlane)
                        uint32_t cc = a.ccode[j];
#if defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("" : "+s"(cc));
#endif
                        const int src = (int)(cc & 15u);
                        if ((FEAT & 1) && src == SRC_SINE) {
                            KEEP_BRANCH();
                            // (the full record of a sine_dd column is read where it is used: zoff is 0, but only known at run time)
                            if constexpr ((FEAT & 1) != 0) { const ColOp co = a.cols[KMAX > 2 ? j + zoff : j]; sine_column(co, x); }
                        } else if constexpr (STAT == 1) {
                            // mean | sum: s / n with n = 2 or 4 is s * (1 / n) exactly; one multiply by a scalar either way (n = 3: the
                            // correctly rounded quotient, already in mean[])
                            const double sc = src == SRC_SUM ? 1.0 : 1.0 / GL;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) x[i] = ((GL == 3 || RAG) && src != SRC_SUM) ? mean[i] : s[i] * sc;
                        } else {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                x[i] = mean[i];                                        // SRC_MEAN: (u + v) / 2, the reference's s / n
                                if (src == SRC_SUM) x[i] = s[i];
                                if (STAT >= 2 && src == SRC_MIN) x[i] = mn[i];
                                if (STAT >= 2 && src == SRC_MAX) x[i] = mx[i];
                            }
                        }
                        // (continuing ONE double-double chain through consecutive exponents of a polynomial — 3 steps instead of 6 for
                        // power[1..4] — was built and measured: the chain state's registers cost more than the steps save, 5.8 -> 6.5 ms
                        // on 1801 x 3600 f32; profiles/r03_pairs_mean_poly.txt)
                        // powi_dd_vec's chain for exponents >= 2 (the host keeps e < 1 off the lean forms; e = 1 adds x itself),
                        // written out: nested scalar branches with straight-line code for the squares, cubes and fourth powers of a
                        // polynomial, and the column's add at every leaf (one merged add would cost a register copy per path)
                        const int n = ((cc >> 4) & 15u) == (uint32_t)TF_POWI ? (int)(cc >> 8) : 1;
                        if (n >= 2) {
                            KEEP_BRANCH();
                            double hi[VEC], lo[VEC];
#pragma unroll
                            for (int i = 0; i < VEC; ++i) hi[i] = x[i] * x[i];
                            if (n >= 3) {
                                KEEP_BRANCH();
                                auto step = [&]() {
#pragma unroll
                                    for (int i = 0; i < VEC; ++i) {
                                        const double p = hi[i] * x[i];
                                        const double err = __fma_rn(hi[i], x[i], -p);
                                        lo[i] = __fma_rn(lo[i], x[i], err);
                                        hi[i] = p;
                                    }
                                };
#pragma unroll
                                for (int i = 0; i < VEC; ++i) lo[i] = __fma_rn(x[i], x[i], -hi[i]);
                                step();
                                if (n >= 4) {
                                    KEEP_BRANCH();
                                    step();
                                    if (n > 4) {
                                        KEEP_BRANCH();
                                        for (int it = 4; it < n; ++it) step();
                                    }
                                }
#pragma unroll
                                for (int i = 0; i < VEC; ++i) os[j][i] += hi[i] + lo[i];
                                asm volatile("; leaf: x^n, n >= 3");          // (distinct tails: the leaves are not merged back)
                            } else {
#pragma unroll
                                for (int i = 0; i < VEC; ++i) os[j][i] += hi[i];
                                asm volatile("; leaf: x^2");
                            }
                            continue;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) os[j][i] += x[i];
                }
            }
            if (emit_slot) {
                if (RF && a.rf_w != nullptr) {
                    KEEP_BRANCH();
                    double val[KMAX][VEC];
#pragma unroll
                    for (int j = 0; j < KMAX; ++j)
#pragma unroll
                        for (int i = 0; i < VEC; ++i) val[j][i] = ((nanacc[i] >> lane) & 1ull) ? nan64() : (LEAN_SINE ? os[j][i] * 0.5 : os[j][i]);
                    if constexpr (RF) rf_emit(val, slot);
                } else {
#pragma unroll
                for (int j = 0; j < KMAX; ++j) {
                    if (j < K) {
                        double* dst = a.partial + ((int64_t)slot * K + j) * C + c0;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const double val = ((nanacc[i] >> lane) & 1ull) ? nan64() : (LEAN_SINE ? os[j][i] * 0.5 : os[j][i]);
                            if (active) dst[i] = val;
                        }
                    }
                }
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) nanacc[i] = 0ull;
                ++slot;
                reset_outer();
            }
            return;
        }
        uint64_t pk[VEC][4];
        if constexpr (SL && TKI) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u) pk[i][u] = 0ull;
        }
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (j < K) {
                // zoff is 0, but only known at run time (it comes from the group table word): the column
                // record is then read from the kernel-argument segment HERE, by scalar loads, instead of
                // being hoisted out of the time loop into ~18 SGPRs per column — which overflowed the
                // SGPR file and came back as v_readlane (VALU) traffic in every group end.
                const ColOp co = a.cols[j + zoff];
                // Every test on a column field below is wave-uniform.  KEEP_BRANCH() pins those tests
                // as real scalar branches: left alone, hipcc if-converts the cheap-looking arms (a
                // float round trip, 1/x for negative exponents) and runs them on every group end.
                double x[VEC];
                const int src = co.src;
                if ((FEAT & 1) && STAT >= 2 && src == SRC_SINE) {
                    KEEP_BRANCH();
                    if constexpr ((FEAT & 1) && STAT >= 2) sine_column(co, x);
                } else
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const bool bad = hasnan[i] || empty;
                    x[i] = nan64();
                    if (STAT >= 1 && src == SRC_MEAN) x[i] = bad ? nan64() : mean[i];
                    else if (STAT >= 1 && src == SRC_SUM) x[i] = bad ? nan64() : s[i];
                    else if (STAT >= 2 && src == SRC_MIN) x[i] = bad ? nan64() : mn[i];
                    else if (STAT >= 2 && src == SRC_MAX) x[i] = bad ? nan64() : mx[i];
                    else if (STAT == 3 && src == SRC_NANMEAN)
                        x[i] = (empty || cnt[i] == 0) ? nan64() : s[i] / (double)cnt[i];
                    else if (NTHR > 0 && src == SRC_THR) {
                        double t = 0.0;
                        bool poisons = false;
#pragma unroll
                        for (int q = 0; q < NTHR; ++q)
                            if (q == co.src_idx) {
                                if constexpr (HB) t = (double)hcnt[((a.hb_bin_of_slot[q] + 1) * VEC + i) * bd + tid];
                                else t = TKI ? (double)cthr[q][i] : acc[q][i];
                                poisons = a.thr[q].nan_poisons != 0;
                            }
                        x[i] = (empty || (poisons && hasnan[i])) ? nan64() : t;
                    }
                }
                if (co.rounding & 1) {                                // the reference stored this step in float32
                    KEEP_BRANCH();
#pragma unroll
                    for (int i = 0; i < VEC; ++i) x[i] = (double)(float)x[i];
                }
                const int tf = co.tf;
                if (tf == TF_POWI) {
                    powi_dd_vec<VEC>(x, co.tf_iarg);
                } else if (tf == TF_HINGE) {
                    KEEP_BRANCH();
                    if (co.rounding & 2) {
                        const float kf = (float)co.tf_arg;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const float xf = (float)x[i];
                            x[i] = (double)(((xf > kf) ? 1.0f : 0.0f) * (xf - kf));
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) x[i] = ((x[i] > co.tf_arg) ? 1.0 : 0.0) * (x[i] - co.tf_arg);
                    }
                } else if ((FEAT & 2) && tf == TF_POW) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) x[i] = pow(x[i], co.tf_arg);
                } else if ((FEAT & 2) && tf == TF_INTER) {
                    // np.multiply(block, other) (dataset.py:563): this group's value times other[g][cell]; lanes beyond the
                    // grid re-read valid cells (c_ld) and are never stored
                    KEEP_BRANCH();
                    const int64_t at = (int64_t)gidx * C + c_ld;
                    if (co.inter_f32) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) x[i] *= (double)((const float*)co.inter)[at + i];
                    } else {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) x[i] *= ((const double*)co.inter)[at + i];
                    }
                    if (co.rounding & 2) {                            // float32 times float32 stays float32 in the reference
#pragma unroll
                        for (int i = 0; i < VEC; ++i) x[i] = (double)(float)x[i];
                    }
                }
                if constexpr (SL && TKI) {
                    if (a.packed) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const uint64_t u = (x[i] != x[i]) ? (uint64_t)a.pk_mask : (uint64_t)(uint32_t)x[i];
                            const int wi = a.pk_word[j];
                            const uint64_t f = u << a.pk_shift[j];
#pragma unroll
                            for (int w = 0; w < 4; ++w) pk[i][w] |= (wi == w) ? f : 0ull;
                        }
                        continue;
                    }
                }
                if constexpr (SL) {
                    if (active) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) a.partial[((int64_t)slot * K + j) * C + c0 + i] = x[i];
                    }
                    continue;
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const double xi = x[i];
                    double o = os[j][i];
                    switch (co.outer) {
                        case OUT_FIRST: o = xi; break;
                        case OUT_SUM:
                        case OUT_MEAN: o += xi; break;              // NaN is sticky
                        case OUT_MIN: { double t = (xi < o) ? xi : o; o = (xi != xi) ? xi : t; break; }
                        case OUT_MAX: { double t = (xi > o) ? xi : o; o = (xi != xi) ? xi : t; break; }
                        case OUT_DD: {
                            const bool m = (xi > co.o0) && (xi < co.o1);
                            o += (xi != xi) ? xi : (m ? fabs(xi - co.obase) : 0.0);
                            break;
                        }
                        default: {  // OUT_BINS: a NaN value is simply out of range
                            const bool m = (xi > co.o0) && (xi < co.o1);
                            o += m ? 1.0 : 0.0;
                            break;
                        }
                    }
                    os[j][i] = o;
                }
            }
        }
        if constexpr (SL && TKI) {
            if (a.packed && active) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const u4 lo = u4{(uint32_t)pk[i][0], (uint32_t)(pk[i][0] >> 32), (uint32_t)pk[i][1], (uint32_t)(pk[i][1] >> 32)};
                    if (a.pk_nw == 2) {
                        // non-temporal stores: the records are not read again by this kernel; measured 1.9 % faster
                        // than plain stores on configs[3] (sc0 sc1 stores 1.6 %), profiles/r01_ab_packed_store_policy.txt
                        __builtin_nontemporal_store(lo, (u4*)((char*)a.partial + ((int64_t)slot * C + c0 + i) * 16));
                    } else {
                        u4* dst = (u4*)((char*)a.partial + ((int64_t)slot * C + c0 + i) * 32);
                        __builtin_nontemporal_store(lo, dst);
                        __builtin_nontemporal_store(u4{(uint32_t)pk[i][2], (uint32_t)(pk[i][2] >> 32), (uint32_t)pk[i][3], (uint32_t)(pk[i][3] >> 32)}, dst + 1);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s[i] = 0.0; mn[i] = inf64(); mx[i] = -inf64(); cnt[i] = 0; nanmask[i] = 0ull;
#pragma unroll
            for (int j = 0; j < NTHR; ++j) { if (HB) {} else if (TKI) cthr[j][i] = 0; else acc[j][i] = 0.0; }
        }
        if constexpr (HB) {
            for (int b = 0; b < hb_bins * VEC; ++b) hcnt[b * bd + tid] = 0;
        }
        if constexpr (SL) {
            ++slot;
        } else if (emit_slot) {
            if (RF && a.rf_w != nullptr) {
                KEEP_BRANCH();
                if constexpr (RF) rf_emit(os, slot);
            } else if (active) {
#pragma unroll
                for (int j = 0; j < KMAX; ++j) {
                    if (j < K) {
                        double* dst = a.partial + ((int64_t)slot * K + j) * C + c0;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) dst[i] = os[j][i];
                    }
                }
            }
            ++slot;
            reset_outer();
        }
    };

    int kk = 0;                 // row index relative to the chunk
    int g = g_lo;

    if constexpr (PAIR) {
        const TIn* p = cube;
        constexpr int GB = DEPTH / GL;                // groups per block of rows
        // the group's statistics: min / max in the input precision (exact), the sum in time order — for a pair min + max, the same
        // two addends the reference adds; a NaN in any row marks the lane (the group is NaN, nb_kernels.py:145-147)
        auto short_stats = [&](const RawVec<TIn, VEC>* r) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const TIn u = r[0].v[i], v = r[1].v[i];
                pnan[i] = u != u || v != v;
                if constexpr (GL == 2 && LEAN_SINE) {
                    // every column a plain sine_dd: the lean group end reads only the pair's sum and |difference| (sine_column) —
                    // no min, no max; mn / mx hold the pair as it came
                    mn[i] = (double)u; mx[i] = (double)v;
                    s[i] = mn[i] + mx[i];
                    continue;
                }
                // v_min / v_max straight on the loaded values: the builtins first canonicalise both operands (v_max x, x)
                // against signalling NaNs; a group with any NaN is a NaN group anyway
                TIn lo, hi;
                if constexpr (sizeof(TIn) == 4) {
                    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(u), "v"(v));
                    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(u), "v"(v));
                } else {
                    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(u), "v"(v));
                    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(u), "v"(v));
                }
                if constexpr (GL == 2) {
                    plo[i] = lo; phi[i] = hi;
                    mn[i] = (double)lo; mx[i] = (double)hi;
                    s[i] = mn[i] + mx[i];
                    if constexpr (STAT == 1) s[i] = (double)u + (double)v;     // (the same two addends; no min / max needed)
                } else if constexpr (GL == 3) {
                    const TIn u2 = r[2].v[i];
                    pnan[i] = pnan[i] || u2 != u2;
                    s[i] = ((double)u + (double)v) + (double)u2;                            // nb_kernels.py:130-137: k ascending
                    if constexpr (STAT >= 2) {
                        if constexpr (sizeof(TIn) == 4) {
                            asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(u2));
                            asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(u2));
                        } else {
                            asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(u2));
                            asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(u2));
                        }
                        plo[i] = lo; phi[i] = hi;
                        mn[i] = (double)lo; mx[i] = (double)hi;
                    }
                } else {
                    const TIn u2 = r[2].v[i], u3 = r[3].v[i];
                    pnan[i] = pnan[i] || u2 != u2 || u3 != u3;
                    s[i] = (((double)u + (double)v) + (double)u2) + (double)u3;             // nb_kernels.py:130-137: k ascending
                    if constexpr (STAT >= 2) {
                        TIn lo2, hi2;
                        if constexpr (sizeof(TIn) == 4) {
                            asm("v_min_f32 %0, %1, %2" : "=v"(lo2) : "v"(u2), "v"(u3));
                            asm("v_max_f32 %0, %1, %2" : "=v"(hi2) : "v"(u2), "v"(u3));
                            asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(lo2));
                            asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(hi2));
                        } else {
                            asm("v_min_f64 %0, %1, %2" : "=v"(lo2) : "v"(u2), "v"(u3));
                            asm("v_max_f64 %0, %1, %2" : "=v"(hi2) : "v"(u2), "v"(u3));
                            asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(lo2));
                            asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(hi2));
                        }
                        plo[i] = lo; phi[i] = hi;
                        mn[i] = (double)lo; mx[i] = (double)hi;
                    }
                }
            }
        };
        // Loads run one block ahead, group by group: as soon as a group's rows have been reduced to its statistics the same
        // group's rows of the NEXT block are requested, so the memory pipe works while the arcs run (a block that loads, then
        // waits, then computes left it idle: VALU busy 0.81 of the kernel; with this 0.90).  The requests sit under the branch
        // "the next block has this group", which makes the compiler land them in fresh registers and move them into place at the
        // loop's back edge behind one s_waitcnt vmcnt(0).  The textbook form — unconditional loads straight into the ring's
        // registers, counted waits vmcnt(DEPTH - GL) per group, no moves — was built too and measured BEHIND this one on every
        // short-group shape (C5 3.59 vs 3.53 ms, mean -> power[1..4] -> sum f32 5.66 vs 5.53, f64 3.74 vs 3.61; the old
        // load-wait-compute block on the latter two: 5.76 / 3.90; profiles/r03_short_group_loads.txt); depths 4, 6, 8 measure alike.
        // Addresses: ONE scalar row pointer that advances by a row per load, in a buffer descriptor, + the lane's 32-bit byte
        // offset — no vector address arithmetic (a global_load needs a v_lshl_add_u64 per row) and two scalar adds per row; the host
        // keeps plans whose rows reach 4 GiB off this path.
        const char* nx = (const char*)a.cube + (size_t)(k_lo * C) * sizeof(TIn);       // the next row to load (uniform)
        const uint32_t voff = (uint32_t)((uint64_t)c_ld * sizeof(TIn));
        const size_t rowb = (size_t)C * sizeof(TIn);
        if constexpr (RAG) {
            // groups of one to four rows: slot q of the block holds group g + q in its four row registers, as many of them filled
            // as the group is long.  Rows are requested in time order through the one scalar row pointer, a group's successor in
            // its slot (group g + GB + q) as soon as its own rows have been reduced — the uniform forms' schedule with a scalar
            // trip count on the loads.
            auto rag_stats = [&](const RawVec<TIn, VEC>* rr, int len) {
                TIn lo[VEC], hi[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const TIn u = rr[0].v[i];
                    pnan[i] = u != u; s[i] = (double)u; lo[i] = u; hi[i] = u;
                }
#pragma unroll
                for (int d = 1; d < GL; ++d) {
                    if (d < len) {
                        KEEP_BRANCH();
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const TIn v = rr[d].v[i];
                            pnan[i] = pnan[i] || v != v;
                            s[i] += (double)v;                                      // nb_kernels.py:130-137: k ascending
                            if constexpr (STAT >= 2) {
                                if constexpr (sizeof(TIn) == 4) {
                                    asm("v_min_f32 %0, %1, %2" : "=v"(lo[i]) : "v"(lo[i]), "v"(v));
                                    asm("v_max_f32 %0, %1, %2" : "=v"(hi[i]) : "v"(hi[i]), "v"(v));
                                } else {
                                    asm("v_min_f64 %0, %1, %2" : "=v"(lo[i]) : "v"(lo[i]), "v"(v));
                                    asm("v_max_f64 %0, %1, %2" : "=v"(hi[i]) : "v"(hi[i]), "v"(v));
                                }
                            }
                        }
                    }
                }
                if constexpr (STAT >= 2) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { plo[i] = lo[i]; phi[i] = hi[i]; mn[i] = (double)lo[i]; mx[i] = (double)hi[i]; }
                }
            };
            RawVec<TIn, VEC> r[DEPTH];
            int lens[GB];
            int k_next = (int)k_lo;                     // the next row to request (the cube's row index)
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                lens[q] = 0;
                if (g + q < g_hi) {
                    const int e = (int)(ld_uniform(&a.gtab[2 * (g + q)]) >> 1) & 0x7fffffff;
                    lens[q] = e - k_next;
                    k_next = e;
                }
#pragma unroll
                for (int d = 0; d < GL; ++d) {
                    if (d < lens[q]) { r[GL * q + d] = ld_stream_row<TIn, VEC, AUX>(nx, voff); nx += rowb; }
                    else r[GL * q + d] = r[0];         // (never read: the statistics stop at the group's length)
                }
            }
            while (g < g_hi) {
                const int ng = (g_hi - g) < GB ? (g_hi - g) : GB;
#pragma unroll
                for (int q = 0; q < GB; ++q) {
                    if (q < ng) {
                        const int64_t w = ld_uniform(&a.gtab[2 * (g + q)]);
                        const int len = lens[q];
                        rag_stats(&r[GL * q], len);
                        if (g + GB + q < g_hi) {
                            const int e = (int)(ld_uniform(&a.gtab[2 * (g + GB + q)]) >> 1) & 0x7fffffff;
                            const int ln = e - k_next;
                            k_next = e;
                            lens[q] = ln;
#pragma unroll
                            for (int d = 0; d < GL; ++d) {
                                if (d < ln) { r[GL * q + d] = ld_stream_row<TIn, VEC, AUX>(nx, voff); nx += rowb; }
                            }
                        }
                        const double inv = len == 3 ? 1.0 / 3.0 : (len == 2 ? 0.5 : (len == 4 ? 0.25 : 1.0));
                        group_end((w & 1) != 0, len, inv, (int)((uint64_t)w >> 63), g + q);
                    }
                }
                g += ng;
            }
        } else {
        RawVec<TIn, VEC> r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (d < rows) { r[d] = ld_stream_row<TIn, VEC, AUX>(nx, voff); nx += rowb; }
            else r[d] = r[0];                      // (a chunk shorter than a block: these rows' groups are not evaluated)
        }
        while (g < g_hi) {
            const int ng = (g_hi - g) < GB ? (g_hi - g) : GB;
            // one copy of the group end per group of the block: since the arcs come from a table the copies fit the instruction
            // cache, and the rows need not be shifted down GL registers per group (round 2's rolled loop: 4 VALU per cell-day)
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                if (q < ng) {
                    const int64_t w = ld_uniform(&a.gtab[2 * (g + q)]);
                    short_stats(&r[GL * q]);
                    if (kk + DEPTH + GL * (q + 1) <= rows) {       // (groups are whole: all of the next block's group q, or none)
#pragma unroll
                        for (int d = 0; d < GL; ++d) { r[GL * q + d] = ld_stream_row<TIn, VEC, AUX>(nx, voff); nx += rowb; }
                    }
                    group_end((w & 1) != 0, GL, 1.0 / GL, (int)((uint64_t)w >> 63), g + q);
                }
            }
            g += ng;
            kk += GL * ng;
        }
        }
    } else if constexpr (PIPE == 0) {
        const TIn* p = cube;
        // histogram variants with arithmetic edges are issue-bound beside their stream: their rows are addressed like the short-group
        // forms' — ONE scalar row pointer in a buffer descriptor + the lane's 32-bit byte offset, no vector address arithmetic (a
        // global_load wants a 64-bit vector address: one v_lshl_add_u64 per row and lane, a fifteenth of the element's instructions);
        // the host keeps plans whose rows reach 4 GiB off these variants
        const char* nx = (const char*)a.cube + (size_t)(k_lo * C) * sizeof(TIn);
        const uint32_t voff = (uint32_t)((uint64_t)c_ld * sizeof(TIn));
        const size_t rowb = (size_t)C * sizeof(TIn);
        // group table word for g is fetched one group ahead: its scalar-load latency hides behind
        // the previous group's work (matters for 1-2 step groups: daily data, tmin/tmax pairs)
        int64_t w_next = ld_uniform(&a.gtab[2 * g]), iv_next = ld_uniform(&a.gtab[2 * g + 1]);
        while (g < g_hi) {
            const int64_t w = w_next, iv = iv_next;
            w_next = ld_uniform(&a.gtab[2 * g + 2]);             // table is padded by one entry
            iv_next = ld_uniform(&a.gtab[2 * g + 3]);
            const int gend = (int)((w >> 1) - k_lo);
            const int gbeg = kk;
            // DEPTH rows in flight per lane inside a group.  (Issuing the next block's loads before the
            // current block is consumed was built and measured: no gain on any shape, -11 % on the f32
            // K = 5 plan — other waves already cover the gap, the extra live registers cost more.  A register
            // ring that re-arms each row's load right after the row is consumed, across group ends, was
            // measured too: f64 6.88 -> 6.48 TB/s, f32 unchanged.  Bursts of DEPTH rows per wave win.)
            auto load_block = [&](RawVec<TIn, VEC> (&r)[DEPTH]) {
                if constexpr (HA) {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) { r[d] = ld_stream_row<TIn, VEC, AUX>(nx, voff); nx += rowb; }
                    return;
                }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) r[d] = ld_stream<TIn, VEC, AUX>(p + (int64_t)d * C);
                p += (int64_t)DEPTH * C;
            };
            auto use_block = [&](const RawVec<TIn, VEC> (&r)[DEPTH]) {
                if constexpr (HA) {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) consume(r[d]);
                } else if constexpr (HB) {
                    int hb_b[DEPTH][VEC];
                    EdgeT hb_e[DEPTH][VEC];
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                        for (int i = 0; i < VEC; ++i) hb_guess(r[d].v[i], hb_b[d][i], hb_e[d][i]);
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        consume(r[d], false);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) hb_count(r[d].v[i], i, hb_b[d][i], hb_e[d][i]);
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) consume(r[d]);
                }
            };
            for (; kk + DEPTH <= gend; kk += DEPTH) {
                RawVec<TIn, VEC> r[DEPTH];
                load_block(r);
                use_block(r);
            }
            // the group's last rows (fewer than DEPTH; ALL rows of a group shorter than DEPTH): their loads
            // are issued together too — one at a time, a 4-step group ran at a quarter of the bandwidth
            if (kk < gend) {
                const int rem = gend - kk;
                RawVec<TIn, VEC> r[DEPTH];
                if constexpr (HA) {
#pragma unroll
                    for (int d = 0; d < DEPTH - 1; ++d)
                        if (d < rem) { r[d] = ld_stream_row<TIn, VEC, AUX>(nx, voff); nx += rowb; }
                } else {
#pragma unroll
                    for (int d = 0; d < DEPTH - 1; ++d)
                        if (d < rem) r[d] = ld_stream<TIn, VEC, AUX>(p + (int64_t)d * C);
                    p += (int64_t)rem * C;
                }
#pragma unroll
                for (int d = 0; d < DEPTH - 1; ++d)
                    if (d < rem) consume(r[d]);
                kk = gend;
            }
            group_end((w & 1) != 0, gend - gbeg, __longlong_as_double(iv), (int)((uint64_t)w >> 63), g);
            ++g;
        }
    } else {
        // ---- LDS-DMA ring, wave-private ----
        unsigned char* ring = dynlds;                                         // waves * DEPTH KiB
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const uint32_t ring_lds = (uint32_t)(uintptr_t)(lds_ptr_t)ring;
        const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)wave * (DEPTH * 1024u));
        const uint32_t rd_lane = wave_lds + (uint32_t)lane * 16u;
        const int last = rows - 1;
        const int64_t row_bytes = C * (int64_t)sizeof(TIn);
        const char* lane_base = (const char*)cube;
        auto issue = [&](int row, int sl) {
            const int r = row < last ? row : last;                       // tail: re-load the last row
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(lane_base + (int64_t)r * row_bytes),
                                             (lds_ptr_t)(uintptr_t)(wave_lds + (uint32_t)sl * 1024u), 16, 0, AUX);
        };
        int sl = 0;  // ring slot of row kk
        if (rows > 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) issue(d, d);
        }
        int64_t w_next = ld_uniform(&a.gtab[2 * g]), iv_next = ld_uniform(&a.gtab[2 * g + 1]);
        while (g < g_hi) {
            const int64_t w = w_next, iv = iv_next;
            w_next = ld_uniform(&a.gtab[2 * g + 2]);
            iv_next = ld_uniform(&a.gtab[2 * g + 3]);
            const int gend = (int)((w >> 1) - k_lo);
            const int gbeg = kk;
            for (; kk < gend; ++kk) {
                u32x4 raw;
                // row kk has landed once at most DEPTH-1 younger DMAs are outstanding
                asm volatile("s_waitcnt vmcnt(%1)\n\t"
                             "ds_read_b128 %0, %2\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(raw)
                             : "n"(DEPTH - 1), "v"(rd_lane + (uint32_t)sl * 1024u)
                             : "memory");
                issue(kk + DEPTH, sl);           // slot is free again: its row is in registers
                sl = (sl + 1 == DEPTH) ? 0 : sl + 1;
                RawVec<TIn, VEC> rv;
                __builtin_memcpy(&rv, &raw, 16);
                consume(rv);
            }
            group_end((w & 1) != 0, gend - gbeg, __longlong_as_double(iv), (int)((uint64_t)w >> 63), g);
            ++g;
        }
        // no DMA may still target this workgroup's LDS when the wave retires
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

}  // namespace afhip
