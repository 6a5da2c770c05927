// afhip_api.hip — C-ABI entry points, plan lowering and kernel dispatch (include/aggfly_hip.h).
//
// Lowering (afhip_plan_create) turns the column list into
//   * one inner accumulator set (STAT mode) + deduplicated threshold slots evaluated on
//     raw data, + one ColOp per column (source, transform, outer reducer);
//   * a chunk table over time: chunks are ranges of whole inner groups; a chunk either
//     holds whole outer periods (each emits its final value) or is a piece of one long
//     period (it emits a partial that k_combine_slots merges in time order);
//   * the kernel variant (dtype, LDS-DMA or direct loads, STAT, slots, columns).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/aggfly_hip.h"
#include "afhip_kernels.h"
#include "afhip_panel_kernels.h"
#include "afhip_lz4_kernels.h"
#include "afhip_variants.h"
#include "afhip_sine_p2_table.h"

using namespace afhip;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(AFHIP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// ---- per-device state ----
// A handle (CSR, plan) belongs to the device that was current when it was created; every entry point that allocates or
// launches for a handle makes that device current for the duration of the call (DeviceGuard) — a caller's worker thread
// starts on device 0 whatever device its parent thread had selected (the reference runs its kernels from dask's pool,
// aggfly/aggregate/nb_kernels.py:271-305), and with one process per GPU on an 8-GPU node rank r's tables must live on card r.
constexpr int MAX_DEVICES = 64;
struct DevState {
    int cus = -1;                  // compute units (hipDeviceProp_t::multiProcessorCount)
    double* sine_tab = nullptr;    // acos table of the sine_dd closed forms (afhip_sine.h: sine_theta), uploaded on first use
    double* sine_p2 = nullptr;     // P2 table of the pair-mode arc (afhip_sine.h: sine_pair_g; afhip_sine_p2_table.h), uploaded on first use
};
DevState g_dev[MAX_DEVICES];
std::mutex g_dev_mu;

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return dev;
}

int cu_count(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return 256;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    DevState& d = g_dev[dev];
    if (d.cus < 0) {
        hipDeviceProp_t p;
        d.cus = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return d.cus;
}

// Makes `dev` the calling thread's current device and puts the previous one back on scope exit.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (dev >= 0 && dev != prev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define GUARD_DEVICE(dev)                                                                                  \
    DeviceGuard guard__(dev);                                                                              \
    if (guard__.err != hipSuccess)                                                                         \
        return fail(AFHIP_E_HIP, "hipSetDevice(%d) failed: %s", (int)(dev), hipGetErrorString(guard__.err))

// Device that owns a device pointer (-1: not a device allocation the runtime knows, e.g. NULL or host memory).
int pointer_device(const void* p) {
    if (!p) return -1;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) return -1;
    return at.device;
}

// The acos table of sine_theta (afhip_sine.h), in the layout the kernel copies into LDS: SINE_ROWS pairs of rows
// (C, S, TH, 0); pair k belongs to phi_k = asin(k / 256):
//   row 2k     (a <= g: theta = pi/2 - asin(u)):  (-cos phi_k, +sin phi_k, pi/2 - phi_k)
//   row 2k + 1 (a >  g: theta = asin(u)):         (+cos phi_k, -sin phi_k, phi_k)
std::vector<double> sine_table_host() {
    std::vector<double> t((size_t)2 * SINE_ROWS * 4, 0.0);
    for (int k = 0; k < SINE_ROWS; ++k) {
        const double sn = std::min(1.0, (double)k / (double)SINE_SCALE);
        const double cs = std::sqrt((1.0 - sn) * (1.0 + sn));
        const double phi = std::asin(sn);
        double* lo = &t[(size_t)(2 * k) * 4];               // the two cases of a k sit next to each other (sine_theta's row address)
        double* hi = &t[(size_t)(2 * k + 1) * 4];
        lo[0] = -cs; lo[1] = sn; lo[2] = 1.57079632679489661923 - phi;
        hi[0] = cs; hi[1] = -sn; hi[2] = phi;
    }
    return t;
}

// pair = true: the P2 table of the pair-mode arc; false: the acos table.  Each lives as long as the process (8 / 11.5 KB per device).
int sine_table_dev(int dev, bool pair, const double** out) {
    *out = nullptr;
    if (dev < 0 || dev >= MAX_DEVICES) return fail(AFHIP_E_INVALID, "device %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_dev_mu);
    DevState& d = g_dev[dev];
    double*& slot = pair ? d.sine_p2 : d.sine_tab;
    if (!slot) {
        std::vector<double> h;
        if (pair) {
            static_assert(AFHIP_SINE_P2_N == SINE_P2_N && SINE_P2_BYTES >= (SINE_P2_N + 1) * 4 * (int)sizeof(double), "table layout");
            h.assign(SINE_P2_BYTES / sizeof(double), 0.0);
            std::copy(afhip_sine_p2_table, afhip_sine_p2_table + (SINE_P2_N + 1) * 4, h.begin());
        } else {
            static_assert(SINE_TAB_BYTES == 2 * SINE_ROWS * 4 * sizeof(double), "table layout");
            h = sine_table_host();
        }
        double* p = nullptr;
        HIP_TRY(hipMalloc((void**)&p, h.size() * sizeof(double)));
        hipError_t e = hipMemcpy(p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(p); return fail(AFHIP_E_HIP, "sine table upload failed: %s", hipGetErrorString(e)); }
        slot = p;
    }
    *out = slot;
    return AFHIP_OK;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int upload(const std::vector<T>& h) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = h.size();
        size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        HIP_TRY(hipMalloc((void**)&p, bytes));
        if (n) HIP_TRY(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
        return AFHIP_OK;
    }
};

}  // namespace

// Long rows (more than four segment lengths) are cut into segments that are summed by separate threads / waves and then added in row order
// (k_csr_combine_segments): real admin-2 tables span four decades of row lengths, and one lane walking a 10^5-entry row
// alone would be the whole kernel's tail.  The segment length follows the table: ~nnz / 4096 entries (64 .. 1024, a
// multiple of 64), so that a table dominated by a few huge rows still spreads over the chip (a 55 k-entry table whose
// longest row holds 19 k: 21.1 ms serial, 1.5 ms in 1024-entry segments, profiles/r02_spmm_skew.txt), while the rows of an
// ordinary table (hundreds of entries) stay whole; a row is cut into at most 128 pieces (the combine adds them serially).
// exact_order plans never segment (table order, one running sum).
constexpr int64_t SPMM_SEG_MIN = 64, SPMM_SEG_MAX = 1024, SPMM_TARGET_SEGS = 4096, SPMM_MAX_PIECES = 128;

struct afhip_csr {
    int device = 0;                // the device the tables live on (current device at afhip_csr_create)
    int64_t R = 0, nnz = 0, n_cells = 0, max_row = 0;
    DevBuf<int64_t> indptr;
    DevBuf<int32_t> cols;
    DevBuf<double> w;
    // work list of the default (non-exact) route: segment v = entries [seg_ptr[v], seg_ptr[v+1]) of one row; its sums go
    // to row seg_dst[v] of the sums buffer: the region's own row (an unsplit region) or scratch row R + k (a piece)
    int64_t nseg = 0, n_extra = 0, n_split = 0;
    DevBuf<int64_t> seg_ptr;
    DevBuf<int32_t> seg_dst;
    DevBuf<int32_t> split_row;      // [n_split] regions that were cut
    DevBuf<int32_t> split_ptr;      // [n_split + 1] their pieces: scratch rows R + [split_ptr[i], split_ptr[i+1])
    // host copies of the table (the run tables of the region-fused period ends are built from them on first use)
    std::vector<int64_t> h_indptr;
    std::vector<int32_t> h_cols;
    std::vector<double> h_w;
    // Region-fused period ends (FusedArgs::rf_w): per wave-tile size (64 * vec cells) the runs of consecutive cells whose e-th table
    // entry (e = 0, 1, in table order) names the same region; the third, fourth ... entries of a cell (junctions of polygons) are kept
    // as "extras" by region.  ok = false: the tables could not be built (no runs at all, or an upload failed).
    struct RfTab {
        bool built = false, ok = false;
        int64_t n_runs = 0;
        DevBuf<double> w2;            // [n_cells][2]
        DevBuf<uint32_t> lane;        // [wave tiles * 64][2] per lane slot: {scan / start / end bits, run indices} (k_fused_temporal: rfbits, rfrid)
        DevBuf<int32_t> tile;         // [wave tiles][2][2]: {first run, need mask} of entry e
        DevBuf<int64_t> reg_ptr;      // [R + 1]
        DevBuf<int32_t> reg_runs;     // [n_runs] run ids by region, ascending inside a region
        // cells in three or more regions: their entries beyond the second ("extras"), by region in table order
        int64_t n_xcells = 0;
        DevBuf<int32_t> xidx;         // [n_cells] index among the cells with extras, -1 = none (empty when the table has none)
        DevBuf<int64_t> xreg_ptr;     // [R + 1]
        DevBuf<int32_t> xcell;        // [extras] index of the entry's cell among the cells with extras
        DevBuf<double> xw;            // [extras]
    };
    RfTab rf[3];                    // vec = 1, 2, 4
    std::mutex rf_mu;
};

// Builds (once) the run tables for wave tiles of 64 * vec cells.  Returns the table, or nullptr when the route does not apply.
static afhip_csr::RfTab* rf_table(afhip_csr* csr, int vec) {
    const int slot = vec == 1 ? 0 : (vec == 2 ? 1 : -1);      // (the twins hold one or two cells per lane)
    if (slot < 0) return nullptr;
    std::lock_guard<std::mutex> lk(csr->rf_mu);
    afhip_csr::RfTab& t = csr->rf[slot];
    if (t.built) return t.ok ? &t : nullptr;
    t.built = true;
    const int64_t C = csr->n_cells, R = csr->R, tc = (int64_t)64 * vec, nt = (C + tc - 1) / tc;
    std::vector<int32_t> reg((size_t)C * 2, -1);
    std::vector<double> w2((size_t)C * 2, 0.0);
    std::vector<int32_t> xidx, xcell;
    std::vector<int64_t> xreg_ptr((size_t)R + 1, 0);
    std::vector<double> xw;
    for (int64_t r = 0; r < R; ++r) {
        for (int64_t j = csr->h_indptr[(size_t)r]; j < csr->h_indptr[(size_t)r + 1]; ++j) {
            const size_t c = (size_t)csr->h_cols[(size_t)j];
            const int e = reg[2 * c] < 0 ? 0 : (reg[2 * c + 1] < 0 ? 1 : 2);
            if (e == 2) {                                            // a third (fourth ...) region on this cell: an extra of region r
                if (xidx.empty()) xidx.assign((size_t)C, -1);
                if (xidx[c] < 0) xidx[c] = (int32_t)t.n_xcells++;
                xcell.push_back(xidx[c]); xw.push_back(csr->h_w[(size_t)j]);
                continue;
            }
            reg[2 * c + e] = (int32_t)r; w2[2 * c + e] = csr->h_w[(size_t)j];
        }
        xreg_ptr[(size_t)r + 1] = (int64_t)xcell.size();
    }
    // (+ 4 empty tiles: the waves of the last workgroup that start beyond the grid look their tile up too)
    // Runs of a tile, numbered in cell order (a run starts where the region changes to one, or at the tile's first cell), and — per lane
    // slot — how the segmented scan of the period end proceeds (k_fused_temporal: rfbits / rfrid / rf_need).  A lane holds `vec`
    // consecutive cells; a cell without an entry e is a stretch of its own, so the scan never adds across the gaps between runs and
    // the steps a wave needs follow its longest RUN.
    std::vector<int32_t> tile((size_t)(nt + 4) * 4, 0), run_region;
    std::vector<uint32_t> lanew((size_t)(nt + 4) * 64 * 2, 0u);
    for (int64_t ti = 0; ti < nt; ++ti)
        for (int e = 0; e < 2; ++e) {
            const int64_t c_lo = ti * tc;
            const int first = (int)run_region.size();
            auto key_of = [&](int64_t c) -> int32_t { return c < C ? reg[(size_t)(2 * c + e)] : -1; };
            bool F[64];
            uint32_t bits[64], rid[64];
            int n_at = 0;
            for (int l = 0; l < 64; ++l) {
                bits[l] = 0; rid[l] = 0; F[l] = false;
                for (int i = 0; i < vec; ++i) {
                    const int64_t c = c_lo + (int64_t)l * vec + i;
                    const int32_t k = key_of(c);
                    const bool first_cell = l == 0 && i == 0, last_cell = l == 63 && i == vec - 1;
                    const bool bnd = first_cell || k != key_of(c - 1) || k < 0;
                    const bool endc = last_cell || key_of(c + 1) != k;
                    if (bnd && k >= 0) { run_region.push_back(k); ++n_at; }
                    rid[l] |= (uint32_t)((n_at - 1) & 0xff) << (8 * i);
                    bits[l] |= (bnd ? 1u : 0u) << (6 + i);
                    bits[l] |= ((endc && k >= 0) ? 1u : 0u) << (8 + i);
                    F[l] = F[l] || bnd;
                }
            }
            int need = n_at > 0 ? 128 : 0;
            // the lane a step reads (k_fused_temporal: dpp64): steps 0 - 3 the lane 1, 2, 4, 8 to the left inside its row of 16; step 4
            // the last lane of the row before (rows 1 and 3); step 5 lane 31 (rows 2 and 3).  (AFHIP_RF_DPP=0 builds: l - 2^st.)
            auto src_of = [](int st, int l) -> int {
#if AFHIP_RF_DPP
                if (st < 4) return (l & 15) >= (1 << st) ? l - (1 << st) : -1;
                if (st == 4) return ((l >> 4) & 1) ? (l & ~15) - 1 : -1;
                return l >= 32 ? 31 : -1;
#else
                return l >= (1 << st) ? l - (1 << st) : -1;
#endif
            };
            for (int st = 0; st < 6; ++st) {
                bool Fn[64];
                for (int l = 0; l < 64; ++l) {
                    const int sl = src_of(st, l);
                    if (sl >= 0 && !F[l]) { bits[l] |= 1u << st; need |= 1 << st; }
                    Fn[l] = F[l] || (sl >= 0 && F[sl]);
                }
                for (int l = 0; l < 64; ++l) F[l] = Fn[l];
            }
            for (int l = 0; l < 64; ++l) {
                if (vec == 2 && ((bits[l] >> 8) & 1u)) need |= 64;
                lanew[(size_t)(ti * 64 + l) * 2] |= bits[l] << (16 * e);
                lanew[(size_t)(ti * 64 + l) * 2 + 1] |= rid[l] << (16 * e);
            }
            tile[(size_t)(ti * 2 + e) * 2] = first;
            tile[(size_t)(ti * 2 + e) * 2 + 1] = need;
        }
    t.n_runs = (int64_t)run_region.size();
    if (t.n_runs == 0 || t.n_runs > INT32_MAX) return nullptr;
    // (no condition on how short the runs are: the route measured ahead down to regions of five cells — monthly f64 3.65 against 3.95 ms
    // with 60,000 regions on 215 x 1440 —; afhip_plan_run only checks that the run sums fit the area of the per-cell values)
    std::vector<int64_t> reg_ptr((size_t)R + 1, 0);
    for (int32_t r : run_region) ++reg_ptr[(size_t)r + 1];
    for (int64_t r = 0; r < R; ++r) reg_ptr[(size_t)r + 1] += reg_ptr[(size_t)r];
    std::vector<int32_t> reg_runs((size_t)t.n_runs);
    { std::vector<int64_t> at(reg_ptr.begin(), reg_ptr.end() - 1);
      for (int64_t q = 0; q < t.n_runs; ++q) reg_runs[(size_t)at[(size_t)run_region[(size_t)q]]++] = (int32_t)q; }
    DeviceGuard g(csr->device);
    if (t.lane.upload(lanew) || t.w2.upload(w2) || t.tile.upload(tile) || t.reg_ptr.upload(reg_ptr) || t.reg_runs.upload(reg_runs)) return nullptr;
    if (t.n_xcells && (t.xidx.upload(xidx) || t.xreg_ptr.upload(xreg_ptr) || t.xcell.upload(xcell) || t.xw.upload(xw))) return nullptr;
    t.ok = true;
    return &t;
}

struct afhip_plan {
    int device = 0;                       // the device the plan's tables and scratch live on (current device at afhip_plan_create)
    bool has_sine = false;                // a column is sine_dd: launches carry the acos table and its LDS
    afhip_plan_desc desc{};
    std::vector<int64_t> ib, ob;          // host copies
    std::vector<afhip_column> columns;
    // lowering
    int stat = 0, nthr = 0, K = 0;
    std::vector<ThrSlot> thr;
    std::vector<ColOp> cols;               // cols[j].inter / inter_f32 are set by afhip_plan_bind_inter
    std::vector<ChunkDesc> chunks;
    std::vector<int32_t> emit;
    std::vector<int64_t> gtab;            // {(end step) << 1 | emit, bits of 1.0/len} per inner group, padded by one
    std::vector<int32_t> slot_ptr;        // [P+1]
    int64_t n_slots = 0;
    const Variant* variant = nullptr;
    const Variant* variant_rf = nullptr;   // its twin with the region-fused period ends compiled in (null: none in the menu)
    bool rf_plan_ok = false;               // the plan's columns and slots allow the route (the table decides the rest at run time)
    int last_route = 0;                    // 1: the last afhip_plan_run took the region-fused route (afhip_plan_describe tells)
    int last_counts_lanes = -1;            // lanes per (row, period) pair of the last run's packed-count gather (1, 4, 8, 16; -1: it did not run)
    int64_t tiles = 0;
    int wg = WG;                          // threads per workgroup (64 for small grids, else 256)
    int hb_n = 0; double hb_c1 = 0, hb_c0 = 0;                          // LDS-histogram bins
    bool hb_arith = false; double hb_w = 0, hb_lo0 = 0, hb_gl = 0, hb_gh = 0, hb_c0b = 0;   // ... with exactly representable edges (+ the biased guess constant)
    int xcd_remap = 1;        // measured +0.2..1 % on configs[1] (profiles/r01_xcd_remap.txt): harmless, kept on
    bool counts_spmm = true;  // packed-count plans: gather the records directly when no per-cell output is asked for
    bool packed = false;      // single-level, all columns plain bin counts: partial holds packed records (FusedArgs::packed)
    PackFmt pk{};             // their format; pk_bw = bits per count
    int hb_bin_of_slot[MAX_THR] = {0};
    double hb_edge[MAX_THR + 1] = {0};
    // device tables
    DevBuf<int64_t> d_ob;
    DevBuf<int64_t> d_gtab;
    DevBuf<ChunkDesc> d_chunks;
    DevBuf<int32_t> d_slot_ptr;
    // workspace
    int64_t ws_partial = 0, ws_panel = 0;   // byte sizes
    void* own_ws = nullptr;                 // plan-owned scratch (callers that hand no workspace): grown by a new hipMalloc, the
    int64_t own_ws_bytes = 0;               // outgrown block is `retired` until the plan is destroyed — no hipFree (a device-wide
    std::vector<void*> retired;             // synchronisation) ever sits on the run path
    double* sums = nullptr;                 // [rows][P][K + 1] of the current run: behind partial + panel in the run's workspace
    int last_ws = 0;                        // 1: the last run used a caller-owned workspace, 2: plan-owned (afhip_plan_describe tells)
    // experiment knobs, read once when the plan is created (never on the run path)
    bool no_slot_spmm = false, no_slots_divide = false, no_counts_divide = false;
    int rf_layout = -1;                               // AFHIP_RF_LAYOUT=slot|run: layout of the run sums forced (rf_run_major)
    int rf_reduce_order = -1;                         // AFHIP_RF_REDUCE_ORDER=r|p: k_rf_reduce's (region, period) pairs region-major / period-major
    int slot_spmm_sub = 0, slot_spmm_order = -1;      // AFHIP_SLOT_SPMM_ORDER=v|p: SlotSpmmArgs::p_major forced off / on
    int counts_spmm_sub = -1;                         // AFHIP_COUNTS_SPMM_SUB=0|4|8|16: lanes per (row, period) pair of the packed-count gather (0: one, table order)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    // per-launch profiling ring (afhip_plan_profile_*): event pairs around the temporal kernel
    std::vector<hipEvent_t> prof_ev;
    int64_t prof_count = 0;
    ~afhip_plan() {      // run with `device` current (afhip_plan_destroy): the members below free their buffers after this body
        if (own_ws) (void)hipFree(own_ws);
        for (void* q : retired) (void)hipFree(q);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        for (auto& e : prof_ev) if (e) (void)hipEventDestroy(e);
    }
};

// ---------------------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------------------
extern "C" const char* afhip_last_error(void) { return g_err.c_str(); }
extern "C" int afhip_abi_version(void) { return AFHIP_ABI_VERSION; }

extern "C" int afhip_build_info(char* buf, int buf_len) {
    int n = 0, arms = 0, rf = 0;
    const Variant* tab = variants_table(&n);
    for (int i = 0; i < n; ++i) { arms += tab[i].production ? 0 : 1; rf += tab[i].rf ? 1 : 0; }
    char tmp[160];
    const int len = snprintf(tmp, sizeof tmp, "menu=%s variants=%d arms=%d region_fused_twins=%d abi=%d", variants_menu(), n, arms, rf, AFHIP_ABI_VERSION);
    if (buf && buf_len > 0) snprintf(buf, buf_len, "%s", tmp);
    return len + 1;
}

extern "C" int afhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int afhip_device_info(int dev, char* name, int name_len, char* arch, int arch_len,
                                 int* n_cus, int64_t* hbm_bytes) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) snprintf(name, name_len, "%s", p.name);
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", p.gcnArchName);
    if (n_cus) *n_cus = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return AFHIP_OK;
}

// ---------------------------------------------------------------------------------------
// CSR
// ---------------------------------------------------------------------------------------
extern "C" int afhip_csr_create(const int64_t* indptr, const int64_t* cols, const double* w,
                                int64_t R, int64_t nnz, int64_t n_cells, afhip_csr** out) {
    if (!out) return fail(AFHIP_E_INVALID, "csr_create: out is NULL");
    *out = nullptr;
    if (R < 0 || nnz < 0 || n_cells <= 0 || !indptr || (nnz && (!cols || !w)))
        return fail(AFHIP_E_INVALID, "csr_create: bad sizes/pointers (R=%lld nnz=%lld n_cells=%lld)",
                    (long long)R, (long long)nnz, (long long)n_cells);
    if (n_cells > INT32_MAX) return fail(AFHIP_E_INVALID, "csr_create: n_cells exceeds int32");
    if (indptr[0] != 0 || indptr[R] != nnz) return fail(AFHIP_E_INVALID, "csr_create: indptr[0] != 0 or indptr[R] != nnz");
    for (int64_t r = 0; r < R; ++r)
        if (indptr[r + 1] < indptr[r]) return fail(AFHIP_E_INVALID, "csr_create: indptr not monotone at row %lld", (long long)r);
    std::vector<int32_t> c32((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) {
        if (cols[j] < 0 || cols[j] >= n_cells)
            return fail(AFHIP_E_INVALID, "csr_create: column %lld out of range at entry %lld", (long long)cols[j], (long long)j);
        c32[(size_t)j] = (int32_t)cols[j];
    }
    auto* h = new afhip_csr();
    h->device = current_device();
    h->R = R; h->nnz = nnz; h->n_cells = n_cells;
    h->h_indptr.assign(indptr, indptr + R + 1); h->h_cols = c32; h->h_w.assign(w, w + nnz);
    std::vector<int64_t> seg_ptr;
    std::vector<int32_t> seg_dst, split_row, split_ptr(1, 0);
    seg_ptr.reserve((size_t)R + 1); seg_dst.reserve((size_t)R);
    auto up64 = [](int64_t x) { return (x + 63) / 64 * 64; };
    int64_t seg = std::min(SPMM_SEG_MAX, std::max(SPMM_SEG_MIN, up64(nnz / SPMM_TARGET_SEGS)));
    if (const char* e = getenv("AFHIP_SPMM_SEG")) seg = std::max<int64_t>(64, up64(atoll(e)));      // experiment knob
    for (int64_t r = 0; r < R; ++r) {
        const int64_t j0 = indptr[r], j1 = indptr[r + 1], len = j1 - j0;
        h->max_row = std::max(h->max_row, len);
        // (rows up to four segments long stay whole: cutting a 150-entry county in two bought nothing and cost every step of an
        // annual panel a segment-merge launch and, with it, the divide fused into the gather)
        if (len <= 4 * seg) {
            seg_ptr.push_back(j0); seg_dst.push_back((int32_t)r);
        } else {
            const int64_t want = std::min(SPMM_MAX_PIECES, (len + seg - 1) / seg);
            const int64_t piece = up64((len + want - 1) / want);
            const int64_t pieces = (len + piece - 1) / piece;
            for (int64_t i = 0; i < pieces; ++i) {
                seg_ptr.push_back(j0 + i * piece);
                seg_dst.push_back((int32_t)(R + h->n_extra + i));
            }
            h->n_extra += pieces;
            split_row.push_back((int32_t)r);
            split_ptr.push_back((int32_t)h->n_extra);
        }
    }
    seg_ptr.push_back(nnz);
    h->nseg = (int64_t)seg_dst.size(); h->n_split = (int64_t)split_row.size();
    if (R + h->n_extra > INT32_MAX) { delete h; return fail(AFHIP_E_INVALID, "csr_create: too many rows"); }
    int rc;
    if ((rc = h->indptr.upload(std::vector<int64_t>(indptr, indptr + R + 1))) ||
        (rc = h->cols.upload(c32)) ||
        (rc = h->w.upload(std::vector<double>(w, w + nnz))) ||
        (rc = h->seg_ptr.upload(seg_ptr)) || (rc = h->seg_dst.upload(seg_dst)) ||
        (rc = h->split_row.upload(split_row)) || (rc = h->split_ptr.upload(split_ptr))) {
        delete h;
        return rc;
    }
    *out = h;
    return AFHIP_OK;
}

extern "C" void afhip_csr_destroy(afhip_csr* csr) {
    if (!csr) return;
    DeviceGuard g(csr->device);
    delete csr;
}

extern "C" int afhip_csr_device(const afhip_csr* csr) { return csr ? csr->device : -1; }

// AGGFLY_HIP_EXACT_ORDER=1 (read once): the standalone spatial entry points then sum in table order like the plans
// created with exact_order.  AFHIP_SPMM_SERIAL=1 (experiment knob): the serial kernel for every route.
static bool env_flag(const char* name) { const char* e = getenv(name); return e && atoi(e) != 0; }
static bool spmm_serial_env() { static const bool v = env_flag("AFHIP_SPMM_SERIAL") || env_flag("AGGFLY_HIP_EXACT_ORDER"); return v; }

static int64_t spmm_rows(const afhip_csr* csr);

// out[r][q] = sum_j w[j] * X[col[j]][q] for every region r (rows [0, R) of `out`, which holds spmm_rows(csr) x Q doubles).
//   exact:  one thread per (r, q) walks the whole row in table order — bit-identical to np.add.at (spatial.py:185);
//   else:   rows in segments of <= SPMM_SEG entries; Q <= 16: one WAVE per segment, lanes stride over the entries and a fixed
//           butterfly adds the 64 partial sums (deterministic, not table order: ~1e-16 from the serial sum); Q > 16: one
//           thread per (segment, q); the pieces of a cut row are then added in row order.
static int launch_spmm(const afhip_csr* csr, const double* X, double* out, int64_t Q, hipStream_t st, bool exact) {
    if (csr->R * Q == 0) return AFHIP_OK;
    if (exact || spmm_serial_env()) {
        const int64_t n = csr->R * Q;
        hipLaunchKernelGGL(k_csr_spmm, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, csr->indptr.p, (const int32_t*)nullptr,
                           csr->cols.p, csr->w.p, X, out, csr->R, Q);
        HIP_TRY(hipGetLastError());
        return AFHIP_OK;
    }
    if (Q <= 16) {
        const unsigned blocks = (unsigned)((csr->nseg + (WG / 64) - 1) / (WG / 64));      // one wave per segment
        if (Q <= 2) hipLaunchKernelGGL(k_csr_spmm_wave<2>, dim3(blocks), dim3(WG), 0, st, csr->seg_ptr.p, csr->seg_dst.p, csr->cols.p, csr->w.p, X, out, csr->nseg, (int)Q);
        else if (Q <= 4) hipLaunchKernelGGL(k_csr_spmm_wave<4>, dim3(blocks), dim3(WG), 0, st, csr->seg_ptr.p, csr->seg_dst.p, csr->cols.p, csr->w.p, X, out, csr->nseg, (int)Q);
        else if (Q <= 8) hipLaunchKernelGGL(k_csr_spmm_wave<8>, dim3(blocks), dim3(WG), 0, st, csr->seg_ptr.p, csr->seg_dst.p, csr->cols.p, csr->w.p, X, out, csr->nseg, (int)Q);
        else hipLaunchKernelGGL(k_csr_spmm_wave<16>, dim3(blocks), dim3(WG), 0, st, csr->seg_ptr.p, csr->seg_dst.p, csr->cols.p, csr->w.p, X, out, csr->nseg, (int)Q);
    } else {
        const int64_t n = csr->nseg * Q;
        hipLaunchKernelGGL(k_csr_spmm, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, csr->seg_ptr.p, csr->seg_dst.p,
                           csr->cols.p, csr->w.p, X, out, csr->nseg, Q);
    }
    HIP_TRY(hipGetLastError());
    if (csr->n_split) {
        const int64_t n = csr->n_split * Q;
        hipLaunchKernelGGL(k_csr_combine_segments, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, out, csr->split_row.p,
                           csr->split_ptr.p, csr->R, Q, csr->n_split);
        HIP_TRY(hipGetLastError());
    }
    return AFHIP_OK;
}

extern "C" int afhip_scatter_block(const afhip_csr* csr, const double* block_dev, int64_t nt,
                                   double* out_dev, void* stream) {
    if (!csr || !block_dev || !out_dev || nt < 0) return fail(AFHIP_E_INVALID, "scatter_block: bad arguments");
    GUARD_DEVICE(csr->device);
    // the drop-in for _scatter_block: always the table-order sum (out_dev holds exactly R rows)
    return launch_spmm(csr, block_dev, out_dev, nt, (hipStream_t)stream, true);
}

extern "C" int afhip_place_box(const void* chunk_dev, void* cube_dev, int elem_size,
                               int64_t by, int64_t bx, int64_t st, int64_t sy, int64_t sx,
                               int64_t nt, int64_t ny, int64_t nx,
                               int64_t NY, int64_t NX, int64_t t0, int64_t y0, int64_t x0, void* stream) {
    if (!chunk_dev || !cube_dev || by <= 0 || bx <= 0 || NY <= 0 || NX <= 0 || nt < 0 || ny < 0 || nx < 0 ||
        st < 0 || sy < 0 || sx < 0 || t0 < 0 || y0 < 0 || x0 < 0 || sy + ny > by || sx + nx > bx || y0 + ny > NY || x0 + nx > NX)
        return fail(AFHIP_E_INVALID, "place_box: box outside the chunk or the cube");
    const int64_t n = nt * ny * nx;
    if (n == 0) return AFHIP_OK;
    if (n > (int64_t)0x7fffffff * WG) return fail(AFHIP_E_INVALID, "place_box: box too large for one launch");
    GUARD_DEVICE(pointer_device(cube_dev));
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + WG - 1) / WG)), block(WG);
    switch (elem_size) {
        case 2: hipLaunchKernelGGL(k_place_box<uint16_t>, grid, block, 0, s, (const uint16_t*)chunk_dev, (uint16_t*)cube_dev, by, bx, st, sy, sx, nt, ny, nx, NY, NX, t0, y0, x0); break;
        case 4: hipLaunchKernelGGL(k_place_box<uint32_t>, grid, block, 0, s, (const uint32_t*)chunk_dev, (uint32_t*)cube_dev, by, bx, st, sy, sx, nt, ny, nx, NY, NX, t0, y0, x0); break;
        case 8: hipLaunchKernelGGL(k_place_box<uint64_t>, grid, block, 0, s, (const uint64_t*)chunk_dev, (uint64_t*)cube_dev, by, bx, st, sy, sx, nt, ny, nx, NY, NX, t0, y0, x0); break;
        default: return fail(AFHIP_E_INVALID, "place_box: elem_size must be 2, 4 or 8");
    }
    HIP_TRY(hipGetLastError());
    return AFHIP_OK;
}

extern "C" int afhip_lz4_decode_streams(const void* comp_dev, const afhip_lz4_stream* streams_dev, int64_t n_streams, int32_t max_dsize,
                                        void* tmp_dev, void* out_dev, int32_t* errors_dev, void* stream) {
    static_assert(sizeof(afhip_lz4_stream) == sizeof(Lz4Stream) && sizeof(afhip_shuffle_block) == sizeof(ShufBlock), "record layouts");
    if (!comp_dev || !streams_dev || !errors_dev || n_streams < 0 || (!tmp_dev && !out_dev))
        return fail(AFHIP_E_INVALID, "lz4_decode_streams: NULL argument");
    if (max_dsize < 0) return fail(AFHIP_E_INVALID, "lz4_decode_streams: negative max_dsize");
    if (n_streams == 0) return AFHIP_OK;
    if (n_streams > 0x7fffffff) return fail(AFHIP_E_INVALID, "lz4_decode_streams: too many streams for one launch");
    GUARD_DEVICE(pointer_device(comp_dev));
    const dim3 g((unsigned)n_streams), b(64);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* c = (const uint8_t*)comp_dev;
    const Lz4Stream* sr = (const Lz4Stream*)streams_dev;
    static const bool prof = [] { const char* e = getenv("AFHIP_LZ4_PROF"); return e && atoi(e) != 0; }();
    if (prof) {      // measuring aid: cycles per phase, summed over the LZ4 streams of this launch, on stderr (synchronous)
        long long* d = nullptr;
        HIP_TRY(hipMalloc(&d, (size_t)n_streams * 8 * sizeof(long long)));
        HIP_TRY(hipMemsetAsync(d, 0, (size_t)n_streams * 8 * sizeof(long long), st));
        hipLaunchKernelGGL((k_lz4_streams_vec<LZ4_NEAR, true>), g, b, 0, st, c, sr, (uint8_t*)tmp_dev, (uint8_t*)out_dev, errors_dev, d);
        std::vector<long long> h((size_t)n_streams * 8);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(d));
        long long tot[8] = {0}, mx = 0;
        for (int64_t i = 0; i < n_streams; ++i) {
            long long sum = 0;
            for (int j = 0; j < 7; ++j) { tot[j] += h[i * 8 + j]; sum += h[i * 8 + j]; }
            tot[7] += h[i * 8 + 7];
            mx = std::max(mx, sum);
        }
        fprintf(stderr, "lz4 prof (shader clock cycles, all streams): parse %lld walk %lld scan %lld owner %lld pending %lld store %lld "
                        "generic %lld; windows %lld; longest stream %lld\n", tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], tot[6], tot[7], mx);
        return AFHIP_OK;
    }
    hipLaunchKernelGGL((k_lz4_streams_vec<LZ4_NEAR, false>), g, b, 0, st, c, sr, (uint8_t*)tmp_dev, (uint8_t*)out_dev, errors_dev, (long long*)nullptr);
    HIP_TRY(hipGetLastError());
    return AFHIP_OK;
}

extern "C" int afhip_unshuffle_blocks(const void* tmp_dev, void* out_dev, const afhip_shuffle_block* blocks_dev, int64_t n_blocks,
                                      int32_t max_bsize, void* stream) {
    if (!tmp_dev || !out_dev || !blocks_dev || n_blocks < 0 || max_bsize < 0) return fail(AFHIP_E_INVALID, "unshuffle_blocks: bad arguments");
    if (n_blocks == 0) return AFHIP_OK;
    if (n_blocks > 65535) return fail(AFHIP_E_INVALID, "unshuffle_blocks: more than 65535 blocks in one call");
    GUARD_DEVICE(pointer_device(out_dev));
    const unsigned tiles = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t)max_bsize / 2 + 255) / 256));
    hipLaunchKernelGGL(k_unshuffle_blocks, dim3(tiles, (unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)tmp_dev,
                       (uint8_t*)out_dev, (const ShufBlock*)blocks_dev);
    HIP_TRY(hipGetLastError());
    return AFHIP_OK;
}

extern "C" int afhip_panel_divide(const double* num_dev, const double* den_dev, double* res_dev, int64_t K, int64_t R,
                                  int64_t P, void* stream) {
    if (!num_dev || !den_dev || !res_dev || K < 0 || R < 0 || P < 0) return fail(AFHIP_E_INVALID, "panel_divide: bad arguments");
    const int64_t n = K * R * P;
    if (n == 0) return AFHIP_OK;
    GUARD_DEVICE(pointer_device(res_dev));
    hipLaunchKernelGGL(k_divide_num_den, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, num_dev, den_dev, res_dev, n, R * P);
    HIP_TRY(hipGetLastError());
    return AFHIP_OK;
}

// The box's streaming-read ceiling for a time-major cube: `launches` back-to-back launches of k_read_probe over [T][row_bytes],
// HIP events around each, one synchronisation at the end.
extern "C" int afhip_read_probe(const void* cube_dev, int64_t T, int64_t row_bytes, int launches, float* ms_out, void* stream) {
    if (!cube_dev || !ms_out || T <= 0 || row_bytes <= 0 || row_bytes % 8 != 0 || launches <= 0 || launches > 1000)
        return fail(AFHIP_E_INVALID, "read_probe: NULL argument, rows that are not multiples of 8 bytes, or a launch count outside 1..1000");
    const int64_t lanes = row_bytes / 8, blocks = (lanes + 63) / 64;
    if (blocks > 0x7fffffff) return fail(AFHIP_E_INVALID, "read_probe: rows too long for one launch");
    GUARD_DEVICE(pointer_device(cube_dev));
    hipStream_t st = (hipStream_t)stream;
    // (a small grid gets time chunks, like the temporal kernel's launches: about 16 single-wave workgroups per CU in all, chunks of 64 rows and more)
    int64_t chunks = 1;
    { const int64_t want = (int64_t)16 * cu_count(pointer_device(cube_dev));
      if (blocks < want) chunks = std::min<int64_t>(std::max<int64_t>(1, T / 64), std::min<int64_t>(65535, (want + blocks - 1) / blocks)); }
    const int64_t rows_per_chunk = (T + chunks - 1) / chunks;
    chunks = (T + rows_per_chunk - 1) / rows_per_chunk;
    uint32_t* out = nullptr;
    HIP_TRY(hipMalloc((void**)&out, (size_t)lanes * (size_t)chunks * sizeof(uint32_t)));
    std::vector<hipEvent_t> ev((size_t)launches + 1, nullptr);
    int rc = AFHIP_OK;
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) { rc = fail(AFHIP_E_HIP, "read_probe: hipEventCreate failed"); break; }
    if (!rc) {
        (void)hipEventRecord(ev[0], st);
        for (int i = 0; i < launches; ++i) {
            hipLaunchKernelGGL(k_read_probe, dim3((unsigned)blocks, (unsigned)chunks), dim3(64), 0, st, (const uint32_t*)cube_dev, row_bytes / 4, T, rows_per_chunk, out);
            (void)hipEventRecord(ev[(size_t)i + 1], st);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipEventSynchronize(ev[(size_t)launches]);
        if (e != hipSuccess) rc = fail(AFHIP_E_HIP, "read_probe: %s", hipGetErrorString(e));
        for (int i = 0; !rc && i < launches; ++i)
            if (hipEventElapsedTime(&ms_out[i], ev[(size_t)i], ev[(size_t)i + 1]) != hipSuccess) rc = fail(AFHIP_E_HIP, "read_probe: event query failed");
    }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    (void)hipFree(out);
    return rc;
}

extern "C" int afhip_transform(const void* x_dev, int x_dtype, int64_t n, int transform, double arg,
                               const void* other_dev, int other_dtype, void* out_dev, int out_dtype, void* stream) {
    if (!x_dev || !out_dev || n < 0) return fail(AFHIP_E_INVALID, "transform: NULL array or negative size");
    if ((x_dtype != AFHIP_F32 && x_dtype != AFHIP_F64) || (out_dtype != AFHIP_F32 && out_dtype != AFHIP_F64))
        return fail(AFHIP_E_INVALID, "transform: dtypes must be AFHIP_F32 or AFHIP_F64");
    TransformArgs ta{};
    ta.x = x_dev; ta.other = nullptr; ta.out = out_dev; ta.n = n;
    ta.x_f32 = x_dtype == AFHIP_F32; ta.out_f32 = out_dtype == AFHIP_F32; ta.other_f32 = 0;
    switch (transform) {
        case AFHIP_TF_POW:
            if (arg == std::floor(arg) && std::fabs(arg) <= 64.0) { ta.tf = TF_POWI; ta.iarg = (int)arg; }
            else { ta.tf = TF_POW; ta.arg = arg; }
            break;
        case AFHIP_TF_HINGE: ta.tf = TF_HINGE; ta.arg = arg; break;
        case AFHIP_TF_INTER:
            if (!other_dev || (other_dtype != AFHIP_F32 && other_dtype != AFHIP_F64))
                return fail(AFHIP_E_INVALID, "transform: AFHIP_TF_INTER needs the second array and its dtype");
            ta.tf = TF_INTER; ta.other = other_dev; ta.other_f32 = other_dtype == AFHIP_F32;
            break;
        default: return fail(AFHIP_E_INVALID, "transform: unknown transform %d", transform);
    }
    if (n == 0) return AFHIP_OK;
    GUARD_DEVICE(pointer_device(out_dev));
    const int64_t per_block = (int64_t)WG * TRANSFORM_PER_THREAD;
    const int64_t blocks = (n + per_block - 1) / per_block;
    if (blocks > 0x7fffffff) return fail(AFHIP_E_INVALID, "transform: array too large for one launch");
    hipLaunchKernelGGL(k_transform, dim3((unsigned)blocks), dim3(WG), 0, (hipStream_t)stream, ta);
    HIP_TRY(hipGetLastError());
    return AFHIP_OK;
}

extern "C" int afhip_spatial_wavg(const afhip_csr* csr, const double* x_dev, int64_t K, int64_t nt,
                                  double* num_dev, double* den_dev, double* res_dev, void* stream) {
    if (!csr || !x_dev || !res_dev || K <= 0 || nt < 0) return fail(AFHIP_E_INVALID, "spatial_wavg: bad arguments");
    if (nt == 0) return AFHIP_OK;
    GUARD_DEVICE(csr->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t C = csr->n_cells, Q = (K + 1) * nt;
    double *panel = nullptr, *sums = nullptr;
    HIP_TRY(hipMallocAsync((void**)&panel, (size_t)(C * Q) * sizeof(double), st));
    hipError_t e = hipMallocAsync((void**)&sums, (size_t)std::max<int64_t>(spmm_rows(csr) * Q, 1) * sizeof(double), st);
    int rc = AFHIP_OK;
    if (e != hipSuccess) {
        rc = fail(AFHIP_E_HIP, "hipMallocAsync failed: %s", hipGetErrorString(e));
    } else {
        const int64_t n = C * nt;
        hipLaunchKernelGGL(k_validity_panel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, x_dev, panel, C, nt, (int)K);
        if ((e = hipGetLastError()) != hipSuccess) rc = fail(AFHIP_E_HIP, "k_validity_panel launch failed: %s", hipGetErrorString(e));
        if (!rc) rc = launch_spmm(csr, panel, sums, Q, st, false);
        const int64_t m = csr->R * nt;                            // one thread per (region, time step)
        if (!rc && m) {
            hipLaunchKernelGGL(k_panel_divide, dim3((unsigned)((m + WG - 1) / WG)), dim3(WG), 0, st, sums, num_dev, den_dev, res_dev, csr->R, nt, (int)K);
            if ((e = hipGetLastError()) != hipSuccess) rc = fail(AFHIP_E_HIP, "k_panel_divide launch failed: %s", hipGetErrorString(e));
        }
    }
    // both scratch buffers go back on every path (stream-ordered: after the kernels that read them)
    (void)hipFreeAsync(panel, st);
    if (sums) (void)hipFreeAsync(sums, st);
    return rc;
}

// ---------------------------------------------------------------------------------------
// plan lowering
// ---------------------------------------------------------------------------------------
static bool is_stat(int c) { return c >= AFHIP_MEAN && c <= AFHIP_NANMEAN; }

static int add_thr_slot(std::vector<ThrSlot>& thr, const double* a3, bool bins) {
    ThrSlot s{};
    s.t0 = a3[0]; s.t1 = a3[1];
    const bool base_is_t0 = (a3[2] == 0.0);                 // nb_kernels.py:167
    if (bins) { s.A = 0.0; s.B = 1.0; }
    else if (base_is_t0) { s.A = 1.0; s.B = -a3[0]; }
    else { s.A = -1.0; s.B = a3[1]; }
    // float thresholds equivalent to the double compares for float inputs
    s.t0f = (float)a3[0]; if ((double)s.t0f > a3[0]) s.t0f = std::nextafterf(s.t0f, -INFINITY);
    s.t1f = (float)a3[1]; if ((double)s.t1f < a3[1]) s.t1f = std::nextafterf(s.t1f, INFINITY);
    s.nan_poisons = bins ? 0 : 1;
    for (size_t i = 0; i < thr.size(); ++i)
        if (!memcmp(&thr[i], &s, sizeof s)) return (int)i;
    thr.push_back(s);
    return (int)thr.size() - 1;
}

static int lower_columns(afhip_plan* pl) {
    const int K = pl->desc.K;
    int stat = 0;
    pl->thr.clear(); pl->cols.clear();
    pl->has_sine = false;
    for (int j = 0; j < K; ++j) {
        const afhip_column& c = pl->columns[j];
        ColOp co{};
        co.rounding = c.rounding;
        switch (c.inner) {
            case AFHIP_MEAN: co.src = SRC_MEAN; stat = std::max(stat, 1); break;
            case AFHIP_SUM: co.src = SRC_SUM; stat = std::max(stat, 1); break;
            case AFHIP_MIN: co.src = SRC_MIN; stat = std::max(stat, 2); break;
            case AFHIP_MAX: co.src = SRC_MAX; stat = std::max(stat, 2); break;
            case AFHIP_NANMEAN: co.src = SRC_NANMEAN; stat = 3; break;
            case AFHIP_DD: co.src = SRC_THR; co.src_idx = add_thr_slot(pl->thr, c.inner_args, false); break;
            case AFHIP_BINS: co.src = SRC_THR; co.src_idx = add_thr_slot(pl->thr, c.inner_args, true); break;
            case AFHIP_SINE_DD:
                co.src = SRC_SINE; stat = std::max(stat, 2);
                co.s0 = c.inner_args[0]; co.s1 = c.inner_args[1];
                co.s0x2 = 2.0 * co.s0; co.s1x2 = 2.0 * co.s1;
                {   // the pair-mode window tests on float data compare in float: s rounded down / up (afhip_kernels.h: ColOp)
                    auto dn = [](double t) { float f = (float)t; return (double)f > t ? std::nextafterf(f, -INFINITY) : f; };
                    auto up = [](double t) { float f = (float)t; return (double)f < t ? std::nextafterf(f, INFINITY) : f; };
                    co.s0dn = dn(co.s0); co.s0up = up(co.s0); co.s1dn = dn(co.s1); co.s1up = up(co.s1);
                }
                co.swidth = co.s1 - co.s0; co.swidth2 = 2.0 * co.swidth;
                pl->has_sine = true;
                if (c.inner_args[2] != 0.0 && c.inner_args[2] != 1.0)
                    return fail(AFHIP_E_INVALID, "column %d: sine_dd flag must be 0 or 1 (temporal.py:324)", j);
                co.skind = (int)c.inner_args[2];
                break;
            default: return fail(AFHIP_E_INVALID, "column %d: unknown inner reducer %d", j, c.inner);
        }
        switch (c.transform) {
            case AFHIP_TF_NONE: co.tf = TF_NONE; break;
            case AFHIP_TF_POW: {
                const double e = c.transform_arg;
                if (e == std::floor(e) && std::fabs(e) <= 64.0) { co.tf = TF_POWI; co.tf_iarg = (int)e; }
                else { co.tf = TF_POW; co.tf_arg = e; }
                break;
            }
            case AFHIP_TF_HINGE: co.tf = TF_HINGE; co.tf_arg = c.transform_arg; break;
            case AFHIP_TF_INTER: co.tf = TF_INTER; break;
            default: return fail(AFHIP_E_INVALID, "column %d: unknown transform %d", j, c.transform);
        }
        // pow() and `inter` are compiled into the all-purpose (STAT 3) variants only (FEAT bit 1)
        if (co.tf == TF_POW || co.tf == TF_INTER) stat = 3;
        switch (c.outer) {
            case AFHIP_IDENTITY: co.outer = OUT_FIRST; break;
            case AFHIP_SUM: co.outer = OUT_SUM; break;
            case AFHIP_MEAN: co.outer = OUT_MEAN; break;
            case AFHIP_MIN: co.outer = OUT_MIN; break;
            case AFHIP_MAX: co.outer = OUT_MAX; break;
            case AFHIP_DD:
            case AFHIP_BINS:
                co.outer = c.outer == AFHIP_DD ? OUT_DD : OUT_BINS;
                co.o0 = c.outer_args[0]; co.o1 = c.outer_args[1];
                co.obase = (c.outer_args[2] == 0.0) ? c.outer_args[0] : c.outer_args[1];
                break;
            default:
                return fail(AFHIP_E_UNSUPPORTED, "column %d: outer reducer %d is not fused (use the staged path)", j, c.outer);
        }
        pl->cols.push_back(co);
    }
    if ((int)pl->thr.size() > MAX_THR) return fail(AFHIP_E_UNSUPPORTED, "more than %d threshold slots in one pass", MAX_THR);
    if (K > MAX_COLS) return fail(AFHIP_E_UNSUPPORTED, "more than %d columns in one pass", MAX_COLS);
    pl->stat = stat;
    pl->nthr = (int)pl->thr.size();
    pl->K = K;
    return AFHIP_OK;
}

// the variant with the region-fused period ends compiled in and every other field equal (null: the menu has none)
static const Variant* twin_of(const Variant* v) {
    int n = 0;
    const Variant* tab = variants_table(&n);
    for (int i = 0; i < n; ++i) {
        const Variant& t = tab[i];
        if (t.rf && t.dtype == v->dtype && t.pipe == v->pipe && t.vec == v->vec && t.stat == v->stat && t.nthr == v->nthr && t.kmax == v->kmax &&
            t.depth == v->depth && t.nt == v->nt && t.tki == v->tki && t.sl == v->sl && t.hb == v->hb && t.ha == v->ha && t.pair == v->pair &&
            t.ss == v->ss && t.quad == v->quad)
            return &t;
    }
    return nullptr;
}

// Chunking.  target_len = time steps a workgroup should stream; a long period is cut on
// inner-group boundaries into pieces (each emits a partial), short consecutive periods are
// packed into one chunk (each emits its own final value).
static int lay_chunks(afhip_plan* pl, int64_t want_chunks);

// dynamic LDS of a launch of the plan's variant with pl->wg threads per workgroup
static size_t plan_lds_bytes(const afhip_plan* pl) {
    size_t lds = pl->variant->pipe == 1 ? (size_t)(pl->wg / 64) * pl->variant->depth * 1024 : 0;
    if (pl->has_sine) lds += (pl->variant->pair && !pl->variant->quad) ? SINE_P2_BYTES : SINE_TAB_BYTES;      // the variant's sine table, behind the ring
    if (pl->variant->hb) lds = (size_t)HB_TABLE_BYTES + (size_t)(pl->hb_n + 2) * pl->variant->vec * pl->wg * 4;
    return lds;
}

// workgroups of the plan's variant one CU holds at once (registers, LDS, wave slots)
static int resident_wgs_per_cu(const afhip_plan* pl) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pl->variant->fn, pl->wg, plan_lds_bytes(pl)) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

static int build_chunks(afhip_plan* pl, int vec) {
    const auto& ib = pl->ib;
    const auto& ob = pl->ob;
    const int64_t G1 = pl->desc.G1, P = pl->desc.P, T = pl->desc.T, C = pl->desc.n_cells;
    // single-wave workgroups when 256-thread tiles cannot give every CU a few workgroups — and on large grids too, unless every
    // workgroup copies a sine table into LDS first (pair-mode sine_dd at 64 threads: 3.23 -> 5.80 ms).  The bare streaming read of
    // this access shape is fastest in single-wave workgroups (scripts/probe/read_bw.hip: 7.02 against 6.69 TB/s at four rows in
    // flight, profiles/r03_read_ceiling.txt) and the plans follow it by less: configs[1] f64 3.142 -> 3.131 ms, f32 1.713 -> 1.693,
    // C1 f32 1.577 -> 1.558, the reference's benchmark shape 5.336 -> 5.281 (same box, arms alternated; 128 threads: 3.234).
    pl->wg = ((C + (int64_t)WG * vec - 1) / ((int64_t)WG * vec) < (int64_t)cu_count(pl->device) || !pl->has_sine) ? 64 : WG;
    // the LDS-histogram kernel: single-wave workgroups and MANY time chunks.  It is short of bytes in flight (waves park 65 % of
    // their cycles on memory at 4.2 waves per SIMD, VALU and LDS far from busy: profiles/r03_c4_bound_pmc.txt), and the more,
    // smaller workgroups the grid offers the fuller the CUs stay: configs[3] f32 3.15 ms (7 chunks of 256 threads) -> 2.84 ms
    // (126 chunks of 64), f64 5.86 -> 5.58 (profiles/r03_sweep_chunks_depth.txt).  Round 1 had measured 4-wave workgroups
    // ahead — at the few chunks of that time.
    const bool hist = pl->variant && pl->variant->hb;
    if (hist) pl->wg = 64;
    if (const char* e = getenv("AFHIP_FORCE_WG")) { int w = atoi(e); if (w == 64 || w == 128 || w == 256) pl->wg = w; }   // experiment knob
    pl->tiles = (C + (int64_t)pl->wg * vec - 1) / ((int64_t)pl->wg * vec);
    // aim for ~4 workgroups per CU over the whole grid, never streaming fewer than 64 steps
    int per_cu = 4;      // measured (profiles/r01_sweep_chunks.txt, r03_sweep_chunks_depth.txt): the fewer time chunks the better once every CU has ~4 workgroups
    if (hist) per_cu = 96;   // ... except for the histogram kernel (above)
    if (const char* e = getenv("AFHIP_WGS_PER_CU")) per_cu = std::max(1, atoi(e));   // experiment knob
    const int64_t want_wgs = (int64_t)cu_count(pl->device) * per_cu * (WG / pl->wg);
    int64_t want_chunks = std::max<int64_t>(1, (want_wgs + pl->tiles - 1) / pl->tiles);
    // Plans with several output periods: up to one time chunk per period.  Cutting ON period boundaries adds no slot and no traffic
    // (the "fewer chunks are better" of round 1 was measured at P = 1, where every cut adds a slot), and the period-end stores are
    // what such plans pay for: with the stores compiled out the configs[1] plan runs P = 12 and P = 73 exactly as fast as P = 1
    // (3.14 ms), with them 3.69 and 4.30 — 150 MB of stores for 0.55 ms, box-dependent (0.22 ms on another box).  The more chunks, the
    // fewer period ends a workgroup carries in the middle of its stream: P = 365 5.24 -> 4.94 ms, P = 73 3.76 -> 3.46, weekly f32
    // 1.72 -> 1.67, the reference's own benchmark shape 5.64 -> 5.53 (profiles/r03_period_end_stores.txt).  lay_chunks still
    // packs periods shorter than 64 steps together; the histogram kernel keeps its own rule.
    // Plans that already get eight chunks or more keep them (configs[2]'s shape, 14 chunks for 40 years: 40 measured 0.5 % behind).
    bool period_chunks = false;
    if (P > 1 && !hist && want_chunks < 8 && !getenv("AFHIP_WGS_PER_CU") && !getenv("AFHIP_NO_PERIOD_CHUNKS")) {
        int64_t wg_cap = 262144;                                    // workgroups a period-chunked launch may have
        if (const char* e = getenv("AFHIP_PERIOD_CHUNK_WGS")) wg_cap = std::max<int64_t>(1024, atoll(e));      // experiment knob
        const int64_t by_period = std::min<int64_t>(P, std::max<int64_t>(1, wg_cap / std::max<int64_t>(pl->tiles, 1)));
        // (a handful of period chunks makes a handful of occupancy rounds with a costly last one: P = 4 measured 2-4 % behind one chunk)
        if (by_period >= 8) { want_chunks = by_period; period_chunks = true; }
    }
    int rc = lay_chunks(pl, want_chunks);
    if (rc) return rc;
    // Rounds.  A CU holds `resident` workgroups of this variant at once; a grid of more workgroups than the chip holds runs in
    // "rounds", and a last round that is mostly empty is paid in full: the reference's own benchmark shape (global 0.25 deg,
    // 2,028 tiles as ONE chunk against 1,536 resident workgroups = 1.32 rounds) ran at 0.65 of the HBM peak, as three chunks
    // (3.96 rounds) at 0.79 (profiles/r03_ref_shape_arms.txt).  When the grid does not fit the chip at once, take the chunk
    // count (of the next few) whose last round is fullest; a grid that fits keeps the fewest chunks, which measured best.
    // (period-aligned chunks are many and short: their last round weighs little, and the search below would cut periods to fill it)
    if (!period_chunks && !getenv("AFHIP_NO_ROUND_FILL") && !getenv("AFHIP_WGS_PER_CU")) {
        const int64_t capacity = (int64_t)resident_wgs_per_cu(pl) * cu_count(pl->device);
        auto fill = [&](int64_t total) { const int64_t rounds = (total + capacity - 1) / capacity; return (double)total / (double)(rounds * capacity); };
        int64_t total = pl->tiles * (int64_t)pl->chunks.size();
        if (capacity > 0 && total > capacity && fill(total) < 0.92) {
            int64_t best_c = want_chunks;
            double best = fill(total);
            size_t last_n = pl->chunks.size();
            for (int64_t c = want_chunks + 1; c <= want_chunks + 12 && best < 0.92; ++c) {
                if ((rc = lay_chunks(pl, c))) return rc;
                if (pl->chunks.size() == last_n) continue;         // (period boundaries: not every count exists)
                last_n = pl->chunks.size();
                const double f = fill(pl->tiles * (int64_t)pl->chunks.size());
                if (f > best + 1e-9) { best = f; best_c = c; }
            }
            if ((rc = lay_chunks(pl, best_c))) return rc;
        }
    }
    return AFHIP_OK;
}

// Chunk table for ~want_chunks time chunks (see build_chunks).
static int lay_chunks(afhip_plan* pl, int64_t want_chunks) {
    const auto& ib = pl->ib;
    const auto& ob = pl->ob;
    const int64_t G1 = pl->desc.G1, P = pl->desc.P, T = pl->desc.T;
    const int64_t target_len = std::max<int64_t>(64, T / std::max<int64_t>(want_chunks, 1));
    // splitting a period adds partial traffic (16 B per extra slot, column and cell, write +
    // read); keep it under ~5 % of the cube: extra_slots*K*16 <= 0.05*T*elem.  (2 % starved the
    // CONUS-window f32 plan of workgroups: 9 chunks 0.229 ms, 22 chunks 0.151 ms.)
    const int64_t elem = pl->desc.dtype == AFHIP_F32 ? 4 : 8;
    double split_frac = 0.05;
    if (const char* e = getenv("AFHIP_SPLIT_FRAC")) split_frac = atof(e);   // experiment knob
    int64_t split_budget = std::max<int64_t>(1, (int64_t)(split_frac * (double)T * (double)elem / (16.0 * std::max(1, pl->K))));
    const bool any_first = std::any_of(pl->cols.begin(), pl->cols.end(), [](const ColOp& c) { return c.outer == OUT_FIRST; });
    const bool may_split = !pl->desc.exact_order && !any_first;

    pl->chunks.clear();
    pl->emit.assign((size_t)std::max<int64_t>(G1, 1), 0);
    pl->slot_ptr.assign((size_t)P + 1, 0);
    int64_t slot = 0;

    auto steps_of = [&](int64_t p) { return ib[(size_t)ob[(size_t)p + 1]] - ib[(size_t)ob[(size_t)p]]; };
    auto groups_of = [&](int64_t p) { return ob[(size_t)p + 1] - ob[(size_t)p]; };
    auto splittable = [&](int64_t p) {
        return may_split && split_budget > 0 && groups_of(p) >= 2 && steps_of(p) >= 2 * target_len;
    };
    auto push_chunk = [&](int64_t g_lo, int64_t g_hi, int64_t slot_base) {
        ChunkDesc c{};
        c.k_lo = ib[(size_t)g_lo]; c.k_hi = ib[(size_t)g_hi];
        c.g_lo = (int32_t)g_lo; c.g_hi = (int32_t)g_hi; c.slot_base = (int32_t)slot_base;
        pl->chunks.push_back(c);
    };

    int64_t p = 0;
    while (p < P) {
        const int64_t g0 = ob[(size_t)p], g1 = ob[(size_t)p + 1];
        if (g1 == g0) {  // empty resample bin: no slot, the combine kernel writes NaN
            pl->slot_ptr[(size_t)p] = (int32_t)slot;
            ++p;
            continue;
        }
        if (splittable(p)) {
            const int64_t steps = steps_of(p);
            const int64_t pieces = std::max<int64_t>(2, std::min<int64_t>({steps / target_len, g1 - g0, split_budget + 1}));
            pl->slot_ptr[(size_t)p] = (int32_t)slot;
            int64_t g = g0, made = 0;
            for (int64_t i = 1; i <= pieces && g < g1; ++i) {
                int64_t ge;
                if (i == pieces) {
                    ge = g1;
                } else {
                    const int64_t k_goal = ib[(size_t)g0] + (steps * i) / pieces;
                    ge = (int64_t)(std::lower_bound(ib.begin() + g + 1, ib.begin() + g1, k_goal) - ib.begin());
                    ge = std::min(ge, g1);
                }
                if (ge <= g) continue;
                push_chunk(g, ge, slot);
                pl->emit[(size_t)ge - 1] = 1;
                ++slot; ++made;
                g = ge;
            }
            split_budget -= std::max<int64_t>(0, made - 1);
            ++p;
            continue;
        }
        // pack whole periods until the chunk holds ~target_len steps
        const int64_t cg0 = g0, slot_base = slot;
        int64_t acc_steps = 0, cg1 = g0;
        bool first = true;
        while (p < P) {
            const int64_t a0 = ob[(size_t)p], a1 = ob[(size_t)p + 1];
            const int64_t st = steps_of(p);
            if (!first && (acc_steps + st > target_len || splittable(p))) break;
            pl->slot_ptr[(size_t)p] = (int32_t)slot;
            if (a1 > a0) { pl->emit[(size_t)a1 - 1] = 1; ++slot; cg1 = a1; }
            acc_steps += st;
            first = false;
            ++p;
        }
        push_chunk(cg0, cg1, slot_base);
    }
    pl->slot_ptr[(size_t)P] = (int32_t)slot;
    pl->n_slots = slot;
    if (pl->chunks.size() > 65535)
        return fail(AFHIP_E_UNSUPPORTED, "plan needs %zu chunks (> 65535 grid.y)", pl->chunks.size());
    return AFHIP_OK;
}

static int validate_desc(const afhip_plan_desc* d) {
    if (!d) return fail(AFHIP_E_INVALID, "plan_create: desc is NULL");
    if (d->T < 0 || d->n_cells <= 0 || d->K <= 0 || d->G1 < 0 || d->P < 0)
        return fail(AFHIP_E_INVALID, "plan_create: bad sizes (T=%lld n_cells=%lld K=%d G1=%lld P=%lld)",
                    (long long)d->T, (long long)d->n_cells, d->K, (long long)d->G1, (long long)d->P);
    if (d->dtype != AFHIP_F32 && d->dtype != AFHIP_F64) return fail(AFHIP_E_INVALID, "plan_create: dtype must be AFHIP_F32 or AFHIP_F64");
    if (!d->inner_bounds || !d->outer_bounds || !d->columns) return fail(AFHIP_E_INVALID, "plan_create: NULL table");
    if (d->inner_bounds[0] != 0 || d->inner_bounds[d->G1] != d->T)
        return fail(AFHIP_E_INVALID, "plan_create: inner_bounds must run from 0 to T");
    for (int64_t g = 0; g < d->G1; ++g)
        if (d->inner_bounds[g + 1] < d->inner_bounds[g]) return fail(AFHIP_E_INVALID, "plan_create: inner_bounds not monotone (time index must be monotonic increasing)");
    if (d->outer_bounds[0] != 0 || d->outer_bounds[d->P] != d->G1)
        return fail(AFHIP_E_INVALID, "plan_create: outer_bounds must run from 0 to G1");
    for (int64_t p = 0; p < d->P; ++p)
        if (d->outer_bounds[p + 1] < d->outer_bounds[p]) return fail(AFHIP_E_INVALID, "plan_create: outer_bounds not monotone");
    if (d->G1 > INT32_MAX - 2) return fail(AFHIP_E_INVALID, "plan_create: too many inner groups");
    return AFHIP_OK;
}

extern "C" int afhip_plan_create(const afhip_plan_desc* desc, afhip_plan** out) {
    if (!out) return fail(AFHIP_E_INVALID, "plan_create: out is NULL");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    auto* pl = new afhip_plan();
    pl->device = current_device();
    pl->desc = *desc;
    pl->ib.assign(desc->inner_bounds, desc->inner_bounds + desc->G1 + 1);
    pl->ob.assign(desc->outer_bounds, desc->outer_bounds + desc->P + 1);
    pl->columns.assign(desc->columns, desc->columns + desc->K);
    pl->desc.inner_bounds = nullptr; pl->desc.outer_bounds = nullptr; pl->desc.columns = nullptr;
    if ((rc = lower_columns(pl))) { delete pl; return rc; }

    // variant: the load path that measured fastest for the dtype and grid size
    // (profiles/r01_sweep_load_arms.txt), subject to row alignment.
    const int64_t C_ = desc->n_cells;
    int want_pipe = 0, want_vec = 1;
    if (desc->dtype == AFHIP_F64) {
        // one cell per lane, direct loads — on small grids too: round 1 had the LDS-DMA ring ahead there (6.1 vs 5.4 TB/s on
        // 104x236), the re-sweep on round 2's kernel has it behind (5.99 vs 6.67 TB/s; profiles/r02_kbench_resweep.txt)
    } else {
        // two cells per lane, unless the plan carries many accumulators (register pressure):
        // one cell per lane measured 1.6x faster on the 13-bin plan (profiles/r01_kbench_c4_f32.json)
        if (C_ % 2 == 0 && pl->nthr < 4 && pl->K < 8) want_vec = 2;
        // ... and unless it is a light one (one or two mean / sum columns, no threshold slots): one cell per lane then keeps
        // more waves resident and measured 6.6 % faster on configs[0] at 215x1440 (6,534 -> 6,966 GB/s), level on the 104x236
        // window (profiles/r02_kbench_light_f32_plans.txt)
        if (pl->stat <= 1 && pl->nthr == 0 && pl->K <= 2) want_vec = 1;
    }
    // short inner groups: the direct path keeps DEPTH rows in flight only INSIDE a group, the LDS-DMA ring
    // prefetches across group ends.  Measured (mean plan, 721x1440 / 1801x3600): 2-step groups f64 4.5 vs
    // 6.0 TB/s, f32 3.5 vs 4.3; 4-step groups f32 4.4 vs 5.6, f64 equal; 8 steps and longer: equal.
    // every inner group exactly two rows ((tmin, tmax) pairs) and min / max / sine columns: the pair-mode variants of the
    // direct-load path keep DEPTH / 2 whole groups in flight, so they need no ring either
    // ... and the same for groups of exactly four rows (6-hourly data), in the lean form only (FEAT bit 10)
    int glen = 0;
    if (desc->G1 > 0 && pl->nthr == 0 && (pl->stat == 1 || pl->stat == 2) && !getenv("AFHIP_NO_PAIR_MODE")) {
        for (int L : {2, 3, 4}) {
            bool all = desc->T == (int64_t)L * desc->G1;
            for (int64_t g = 0; all && g < desc->G1; ++g) all = pl->ib[(size_t)g + 1] - pl->ib[(size_t)g] == L;
            if (all) glen = L;
        }
        // mixed lengths of one to four rows (a sub-daily series with missing steps): the four-row form with a length per group
        // (FEAT bit 13); a series of single rows throughout is not a short-group plan
        if (glen == 0 && desc->T > desc->G1 && !getenv("AFHIP_NO_RAGGED_MODE")) {
            bool all = true;
            for (int64_t g = 0; all && g < desc->G1; ++g) {
                const int64_t L = pl->ib[(size_t)g + 1] - pl->ib[(size_t)g];
                all = L >= 1 && L <= 4;
            }
            if (all) glen = 5;
        }
    }
    if (glen >= 3 && getenv("AFHIP_NO_QUAD_MODE")) glen = 0;
    // (its loads address a row by a 32-bit byte offset per lane: rows of 4 GiB and more take the general path)
    if ((uint64_t)desc->n_cells * (desc->dtype == AFHIP_F64 ? 8u : 4u) >= (1ull << 32)) glen = 0;
    bool pairs = glen >= 2;                              // short-group mode (two-, three- or four-row groups)
    const bool quad_len = glen >= 3;                     // three / four / mixed rows: the lean form only, general sine closed forms
    const int glcode = glen == 5 ? 3 : (glen == 4 ? 1 : (glen == 3 ? 2 : 0));      // Variant::quad
    // pair plans whose columns are all  mean | sum | min | max | sine_dd -> (integer power) -> sum | mean  without float32 rounding
    // take the lean group end (FEAT bit 8); when every column is a plain sine_dd, its tightest form (FEAT bit 9).  A sine_dd
    // column there needs s0 < s1: its two max() terms are one clamp of width s1 - s0.
    bool lean = pairs && !getenv("AFHIP_NO_LEAN_PAIRS"), lean_sine = lean && pl->K <= 2 && !quad_len;
    for (const ColOp& c : pl->cols) {
        const bool sine_ok = c.src == SRC_SINE && c.s0 < c.s1 && std::isfinite(c.swidth);
        const bool src_ok = c.src == SRC_MEAN || c.src == SRC_SUM || c.src == SRC_MIN || c.src == SRC_MAX || sine_ok;
        lean = lean && src_ok && (c.tf == TF_NONE || (c.tf == TF_POWI && c.tf_iarg >= 1)) && c.rounding == 0 && (c.outer == OUT_SUM || c.outer == OUT_MEAN);
        lean_sine = lean_sine && sine_ok && c.tf == TF_NONE;
    }
    lean_sine = lean_sine && lean;
    // four-row groups exist in the lean form only, for as many columns as its variants hold
    if (quad_len && !(lean && find_variant(desc->dtype, 0, pl->stat, 0, pl->K, 0, 0, false, false, false, false, false, 1, 0, glcode)))
        pairs = lean = false;
    // mean / sum columns alone (no min, max or sine): the pair path exists in the lean form only, and a light plan (one or two
    // columns) streams faster through the LDS-DMA ring, whose prefetch runs across the two-row groups (5.99 vs 5.44 TB/s on
    // 1801 x 3600 f32); with more columns the lean group end wins (profiles/r03_pairs_mean_poly.txt)
    // (four-row groups: the lean form measured ahead of the ring at every column count, profiles/r03_quad_groups.txt)
    if (pairs && pl->stat == 1) {
        // (three-row groups on float32: one- and two-column plans stream faster through the ring, 4.59 / 4.94 against 4.93 / 5.04 ms on
        // 1801 x 3600; from three columns on, and on float64 at every count, the lean form is ahead: profiles/r04_three_row_groups.txt)
        // (mixed lengths likewise: 6.25 / 6.39 against 6.89 / 6.83 ms, float64 level; profiles/r04_mixed_short_groups.txt)
        int min_k = quad_len ? (((glen == 3 || glen == 5) && desc->dtype == AFHIP_F32) ? 3 : 1) : 3;
        if (const char* e = getenv("AFHIP_LEAN_STAT1_MIN_K")) min_k = atoi(e);      // experiment knob
        if (!lean || pl->K < min_k) pairs = lean = lean_sine = false;
    }
    if (!pairs) {
        const double avg_group = desc->G1 > 0 ? (double)desc->T / (double)desc->G1 : 0.0;
        const int vec16 = desc->dtype == AFHIP_F64 ? 2 : 4;
        bool sine = false;                          // sine_dd on short windows is fp64-VALU-bound: direct loads measured 4 % faster
        for (const ColOp& c : pl->cols) sine = sine || c.src == SRC_SINE;
        if (!sine && avg_group > 0 && avg_group < (desc->dtype == AFHIP_F64 ? 4.0 : 8.0) && C_ % vec16 == 0) { want_pipe = 1; want_vec = vec16; }
    }
    int tuning = desc->tuning;
    if (tuning > 0) {
        const int tvec = ((tuning % 10000) % 1000) / 100;
        if (tvec <= 0 || C_ % tvec != 0) tuning = 0;          // a vector arm needs rows that are multiples of it
    }
    // specialisations the lowered plan qualifies for
    bool all_bins = pl->nthr > 0;
    for (const ThrSlot& t : pl->thr) all_bins = all_bins && t.nan_poisons == 0;
    bool single_level = desc->P == desc->G1;
    for (int64_t p = 0; single_level && p <= desc->P; ++p) single_level = pl->ob[(size_t)p] == p;
    for (const ColOp& c : pl->cols) single_level = single_level && c.outer == OUT_FIRST;
    // contiguous equal-width partition?  (sorted by t0, t1[b] == t0[b+1], constant width)
    pl->hb_n = 0;
    if (all_bins && pl->nthr >= 4) {
        std::vector<int> order((size_t)pl->nthr);
        for (int i = 0; i < pl->nthr; ++i) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return pl->thr[(size_t)x].t0 < pl->thr[(size_t)y].t0; });
        const double e0 = pl->thr[(size_t)order[0]].t0;
        const double w = pl->thr[(size_t)order[0]].t1 - e0;
        bool ok = w > 0 && std::isfinite(e0) && std::isfinite(w);
        for (int b = 0; ok && b < pl->nthr; ++b) {
            const ThrSlot& t = pl->thr[(size_t)order[(size_t)b]];
            ok = std::fabs(t.t0 - (e0 + b * w)) <= 1e-9 * w && std::fabs(t.t1 - (e0 + (b + 1) * w)) <= 1e-9 * w;
            if (ok && b + 1 < pl->nthr) ok = t.t1 == pl->thr[(size_t)order[(size_t)b + 1]].t0;
        }
        // the in-kernel guess floor(v / w - e0 / w) is computed in the INPUT precision and may be off by
        // one bin at most: the bins must not be narrower than ~2^20 (f32) / 2^48 (f64) ulps of the edges
        const double emax = std::max(std::fabs(e0), std::fabs(e0 + pl->nthr * w));
        const double eps = desc->dtype == AFHIP_F32 ? 1.2e-7 : 2.3e-16;
        ok = ok && emax * eps * 16.0 < w;
        if (ok) {
            pl->hb_n = pl->nthr; pl->hb_c1 = 1.0 / w; pl->hb_c0 = 1.0 - e0 / w;      // + 1: bin 0 is the lower guard bin
            for (int b = 0; b < pl->nthr; ++b) {
                pl->hb_bin_of_slot[order[(size_t)b]] = b;
                pl->hb_edge[b] = pl->thr[(size_t)order[(size_t)b]].t0;
            }
            pl->hb_edge[pl->nthr] = pl->thr[(size_t)order[(size_t)pl->nthr - 1]].t1;
            // arithmetic edges: E[g] = lo0 + g * w must come out EXACTLY, in the input precision and by the very fma the
            // kernel executes, for every bin of the guarded partition; the clamp points must lie inside the guard bins
            const int n = pl->nthr;
            const double lo0 = e0 - w, gl = e0 - 0.5 * w, gh = pl->hb_edge[n] + 0.5 * w;
            bool ex = true;
            if (desc->dtype == AFHIP_F32) {
                const float wf = (float)w, lo0f = (float)lo0, e0f = (float)e0, glf = (float)gl, ghf = (float)gh;
                ex = (double)wf == w && (double)lo0f == lo0 && (double)e0f == e0 && lo0f + wf == e0f;
                for (int g = 0; ex && g <= n + 1; ++g) {
                    const double lo_want = g == 0 ? lo0 : pl->hb_edge[g - 1];
                    const double hi_want = g == n + 1 ? pl->hb_edge[n] + w : pl->hb_edge[g];
                    ex = (double)fmaf((float)g, wf, lo0f) == lo_want && (double)fmaf((float)g, wf, e0f) == hi_want;
                }
                ex = ex && (double)glf > lo0 && (double)glf < e0 && (double)ghf > pl->hb_edge[n] && (double)ghf < pl->hb_edge[n] + w;
            } else {
                ex = lo0 + w == e0;
                for (int g = 0; ex && g <= n + 1; ++g) {
                    const double lo_want = g == 0 ? lo0 : pl->hb_edge[g - 1];
                    const double hi_want = g == n + 1 ? pl->hb_edge[n] + w : pl->hb_edge[g];
                    ex = fma((double)g, w, lo0) == lo_want && fma((double)g, w, e0) == hi_want;
                }
                ex = ex && gl > lo0 && gl < e0 && gh > pl->hb_edge[n] && gh < pl->hb_edge[n] + w;
            }
            // the one-sided guess (ha_update): the guess constant biased down by the smallest delta of a ladder for which, with the
            // kernel's own fma in the input precision, every edge E[k] of the guarded partition guesses bin k - 1 and the clamp
            // points guess their guard bins.  fma and floor are monotone in v, so every value of [E[t], E[t+1]) then guesses t - 1
            // or t.  No delta fits (bins of a few ulps): the table form.
            double c0b = 0.0;
            if (ex) {
                const double c1 = pl->hb_c1, c0 = pl->hb_c0;
                bool found = false;
                for (int k = (desc->dtype == AFHIP_F32 ? 22 : 50); !found && k >= 8; --k) {
                    const double delta = std::ldexp(1.0, -k);
                    bool okd = true;
                    if (desc->dtype == AFHIP_F32) {
                        const float c1f = (float)c1, cbf = (float)(c0 - delta);
                        auto guess = [&](double v) { return (double)floorf(fmaf((float)v, c1f, cbf)); };
                        for (int g = 1; okd && g <= n + 1; ++g) okd = guess(pl->hb_edge[g - 1]) == (double)(g - 1);
                        okd = okd && guess(gl) == 0.0 && guess(gh) == (double)(n + 1);
                        if (okd) c0b = (double)cbf;
                    } else {
                        const double cb = c0 - delta;
                        auto guess = [&](double v) { return floor(fma(v, c1, cb)); };
                        for (int g = 1; okd && g <= n + 1; ++g) okd = guess(pl->hb_edge[g - 1]) == (double)(g - 1);
                        okd = okd && guess(gl) == 0.0 && guess(gh) == (double)(n + 1);
                        if (okd) c0b = cb;
                    }
                    found = okd;
                }
                ex = found;
            }
            // (these variants address a row by a 32-bit byte offset per lane)
            if ((uint64_t)desc->n_cells * (desc->dtype == AFHIP_F64 ? 8u : 4u) >= (1ull << 32)) ex = false;
            if (getenv("AFHIP_NO_ARITH_EDGES")) ex = false;       // experiment knob: force the table form
            pl->hb_arith = ex; pl->hb_w = w; pl->hb_lo0 = lo0; pl->hb_gl = gl; pl->hb_gh = gh; pl->hb_c0b = c0b;
        }
    }
    if (pl->hb_n > 0 && tuning == 0) { want_pipe = 0; want_vec = 1; }   // the LDS histogram lives on the direct-load path
    const bool partition = pl->hb_n > 0 && want_pipe == 0;
    const bool arith = partition && pl->hb_arith;
    pairs = pairs && want_pipe == 0;
    lean = lean && pairs; lean_sine = lean_sine && pairs;
    // rows in flight per lane on the direct-load path: f64 four, f32 eight — but four for f32 plans with two cells per lane on
    // grids large enough for 256-thread workgroups (the multi-column plans; see gen_variants.py)
    int depth_hint = desc->dtype == AFHIP_F64 ? 4 : 8;
    if (desc->dtype == AFHIP_F32 && want_pipe == 0 && want_vec == 2 && !pairs && pl->hb_n == 0 &&
        (C_ + (int64_t)WG * 2 - 1) / ((int64_t)WG * 2) >= (int64_t)cu_count(pl->device))
        depth_hint = 4;
    if (const char* e = getenv("AFHIP_DEPTH_HINT")) depth_hint = atoi(e);      // experiment knob
    const int quads = (pairs && quad_len) ? glcode : 0;
    const bool twos = pairs && !quad_len;
    if (quads) depth_hint = glen == 3 ? 6 : 8;           // two groups per block
    const int lean_code = lean ? (lean_sine ? 2 : 1) : 0;
    const Variant* v = find_variant(desc->dtype, want_pipe, pl->stat, pl->nthr, pl->K, tuning, want_vec, all_bins, single_level, partition, arith, twos, lean_code, depth_hint, quads);
    if (!v && tuning > 0)   // a tuning arm is a hint: arms are compiled for the headline plan shapes only
        v = find_variant(desc->dtype, want_pipe, pl->stat, pl->nthr, pl->K, 0, want_vec, all_bins, single_level, partition, arith, twos, lean_code, depth_hint, quads);
    if (!v) v = find_variant(desc->dtype, 0, pl->stat, pl->nthr, pl->K, 0, 1, all_bins, single_level, pl->hb_n > 0, pl->hb_n > 0 && pl->hb_arith, twos, lean_code, depth_hint, quads);
    if (!v) {
        delete pl;
        return fail(AFHIP_E_UNSUPPORTED, "no kernel variant for dtype=%d stat=%d slots=%d columns=%d", desc->dtype, pl->stat, pl->nthr, pl->K);
    }
    // A single-level plan takes an `sl` variant when the menu has one (no outer accumulators: cheaper) — but those have no region-fused
    // twin, and a plan that stores one value per group, column and cell (a daily panel of several degree-day columns) then writes and
    // re-reads period values worth a sizeable share of the cube.  From 5 % on the general two-level variant (outer = first) with its twin
    // is taken instead; packed bin counts (16-byte records, gathered directly) stay where they are.
    if (v->sl && !v->tki && tuning == 0 && !desc->exact_order && !getenv("AFHIP_NO_REGION_FUSED") &&
        (double)desc->P * pl->K * 8.0 >= 0.05 * (double)desc->T * (desc->dtype == AFHIP_F32 ? 4.0 : 8.0)) {
        const Variant* v2 = find_variant(desc->dtype, want_pipe, pl->stat, pl->nthr, pl->K, 0, want_vec, false, false, false, false, twos, lean_code, depth_hint, quads);
        if (v2 && v2->pipe == 0 && !v2->tki && !v2->hb && twin_of(v2)) v = v2;
    }
    pl->variant = v;
    if ((rc = build_chunks(pl, v->vec))) { delete pl; return rc; }
    // Region-fused period ends (FusedArgs::rf_w): the twin variant, if the menu has one, and what the plan itself must satisfy —
    // two-level columns without float32 rounding of the final value (the period value must enter the weighted sum as it leaves
    // the accumulator; an outer mean's division by the period's group count is applied to the region sums, k_rf_reduce), at most one slot per period (shared validity needs the whole period's value), several periods
    // (with one the stores sit at the kernel's end and cost nothing: the headline stays on the route it was measured on).
    {
        pl->variant_rf = twin_of(v);
        bool ok = pl->variant_rf != nullptr && !desc->exact_order && desc->P >= 2 && !getenv("AFHIP_NO_REGION_FUSED");
        for (const ColOp& c : pl->cols) ok = ok && !(c.rounding & AFHIP_ROUND_FINAL);      // (identity outers too: a daily panel of daily means)
        for (int64_t p = 0; ok && p < desc->P; ++p) ok = pl->slot_ptr[(size_t)p + 1] - pl->slot_ptr[(size_t)p] <= 1;
        // ... and per-cell period values that would be a noticeable share of the traffic: P K 8 bytes per cell against T elem.  Below
        // ~0.2 % there is nothing to win and the emit still costs: configs[2]'s shape (40 annual values of 2 columns from 350,640
        // hourly steps: 0.05 %) measured 0.25 % behind, the one-period headline (0.06 %) 0.6 %; the shapes that gain sit at 0.5 % and up.
        const double share = (double)desc->P * pl->K * 8.0 / std::max(1.0, (double)desc->T * (desc->dtype == AFHIP_F32 ? 4.0 : 8.0));
        const bool forced = getenv("AFHIP_FORCE_REGION_FUSED") != nullptr;
        if (ok && !forced) ok = share >= 0.002;
        // Which forms gain was measured, not derived (profiles/r03_region_fused.txt: an occupancy rule could not tell them apart): float64
        // forms and float32 forms without threshold slots gain 3 - 50 % from two periods on; the lean four-row forms and the six-column
        // lean pair form likewise (6-hourly monthly polynomial: step 0.97 against 1.13 - 1.33 ms).
        // Round 4 (the scan form of the period end; twins for threshold-only plans and every short-group form; profiles/r04_region_fused_scan.txt):
        // the float32-with-a-threshold-slot forms are level from 12 periods and ahead from there, a degree-day column alone gains 6 % at
        // 12 and 52 periods and 22-25 % on a daily panel — one rule for all of them: period values of 0.2 % of the cube and more.  The
        // two-row forms other than the six-column lean one (sine_dd from (tmin, tmax) pairs: a period end every few rows) pay only where
        // the per-cell route's own traffic decides: monthly 4.44 against 3.77 ms (behind), weekly 5.80 against 6.20, daily 17.7 against
        // 21.5 — from period values of 5 % of the cube.
        const bool two_row_light = v->pair && !v->quad && !(v->ss == 1 && v->kmax == 6);
        if (ok && !forced && two_row_light) {
            double min_share = 0.05;
            if (const char* e = getenv("AFHIP_RF_MIN_SHARE")) min_share = atof(e);      // experiment knob
            ok = share >= min_share;
        }
        pl->rf_plan_ok = ok;
    }

    pl->gtab.assign(2 * ((size_t)desc->G1 + 2), 0);
    for (int64_t g = 0; g < desc->G1; ++g) {
        const int64_t len = pl->ib[(size_t)g + 1] - pl->ib[(size_t)g];
        const double inv = len > 0 ? 1.0 / (double)len : 0.0;     // correctly rounded: div_by() then equals s / n exactly
        int64_t bits;
        memcpy(&bits, &inv, 8);
        pl->gtab[2 * (size_t)g] = (pl->ib[(size_t)g + 1] << 1) | (pl->emit[(size_t)g] ? 1 : 0);
        pl->gtab[2 * (size_t)g + 1] = bits;
    }
    if ((rc = pl->d_ob.upload(pl->ob)) || (rc = pl->d_gtab.upload(pl->gtab)) ||
        (rc = pl->d_chunks.upload(pl->chunks)) || (rc = pl->d_slot_ptr.upload(pl->slot_ptr))) {
        delete pl;
        return rc;
    }
    const int64_t C = desc->n_cells, K = desc->K, P = desc->P;
    auto a256 = [](int64_t b) { return (b + 255) / 256 * 256; };
    // packed counts: integer-bin single-level variant, every column a plain bin count, no period longer than a
    // 16-bit counter holds (0xFFFF is the NaN mark)
    // experiment knobs, read once per plan (never on the run path)
    if (const char* e = getenv("AFHIP_XCD_REMAP")) pl->xcd_remap = atoi(e) ? 1 : 0;
    pl->counts_spmm = !getenv("AFHIP_NO_COUNTS_SPMM");
    if (const char* e = getenv("AFHIP_COUNTS_SPMM_SUB")) pl->counts_spmm_sub = atoi(e);
    pl->no_slot_spmm = env_flag("AFHIP_NO_SLOT_SPMM");            // keep k_combine_slots + k_csr_spmm on every route
    pl->no_slots_divide = getenv("AFHIP_NO_SLOTS_DIVIDE") != nullptr;
    pl->no_counts_divide = getenv("AFHIP_NO_COUNTS_DIVIDE") != nullptr;
    if (const char* e = getenv("AFHIP_RF_LAYOUT")) pl->rf_layout = (e[0] == 'r') ? 1 : 0;
    if (const char* e = getenv("AFHIP_RF_REDUCE_ORDER")) pl->rf_reduce_order = (e[0] == 'p') ? 1 : 0;
    if (const char* e = getenv("AFHIP_SLOT_SPMM_ORDER")) pl->slot_spmm_order = (e[0] == 'p') ? 1 : 0;
    if (const char* e = getenv("AFHIP_SLOT_SPMM_SUB")) { const int sb = atoi(e); if (sb == 8 || sb == 16 || sb == 32 || sb == 64) pl->slot_spmm_sub = sb; }
    pl->packed = v->tki && v->sl && K <= 16 && !getenv("AFHIP_NO_PACKED_COUNTS");
    for (const ColOp& c : pl->cols)
        pl->packed = pl->packed && c.src == SRC_THR && c.tf == TF_NONE && c.rounding == 0 && c.outer == OUT_FIRST;
    int64_t maxlen = 0;
    for (int64_t g = 0; g < desc->G1; ++g) maxlen = std::max(maxlen, pl->ib[(size_t)g + 1] - pl->ib[(size_t)g]);
    pl->packed = pl->packed && maxlen < 65535;
    if (pl->packed) {
        // the narrowest field that holds the longest period's count and keeps all ones free for NaN; 16-byte records when
        // K such fields fit two words (daily data, annual bins: 13 x 9 bits), else 16-bit fields in 32 bytes
        int bw = 1;
        while (((int64_t)1 << bw) - 1 <= maxlen) ++bw;
        int f = 64 / bw;
        pl->pk.nw = 2;
        if (K > 2 * f || getenv("AFHIP_PACK32")) { bw = 16; f = 4; pl->pk.nw = 4; }
        pl->pk.mask = (uint32_t)(((uint64_t)1 << bw) - 1);
        for (int j = 0; j < MAX_COLS; ++j) { pl->pk.word[j] = (uint8_t)(j / f); pl->pk.shift[j] = (uint8_t)((j % f) * bw); }
    }
    pl->ws_partial = pl->packed ? a256(std::max<int64_t>(pl->n_slots, 1) * C * pl->pk.nw * 8)
                                : a256(std::max<int64_t>(pl->n_slots, 1) * K * C * 8);
    pl->ws_panel = a256(C * (K + 1) * std::max<int64_t>(P, 1) * 8);
    *out = pl;
    return AFHIP_OK;
}

extern "C" void afhip_plan_destroy(afhip_plan* plan) {
    if (!plan) return;
    DeviceGuard g(plan->device);
    delete plan;
}

extern "C" int afhip_plan_device(const afhip_plan* plan) { return plan ? plan->device : -1; }

// A plan's tables and scratch live on plan->device: a cube (or a second cube, an output, a workspace) on another card would be
// read or written across xGMI at best and, with peer access off, fault.  Refuse it before anything is launched (pointers the
// runtime cannot classify are let through).
static int check_ptr_device(const afhip_plan* pl, const void* ptr_dev, const char* who, const char* what) {
    const int dev = pointer_device(ptr_dev);
    if (dev >= 0 && dev != pl->device)
        return fail(AFHIP_E_INVALID, "%s: %s lives on device %d but the plan was created on device %d "
                    "(create the plan with the data's device current)", who, what, dev, pl->device);
    return AFHIP_OK;
}
static int check_cube_device(const afhip_plan* pl, const void* cube_dev, const char* who) { return check_ptr_device(pl, cube_dev, who, "the cube"); }

extern "C" int afhip_plan_bind_inter(afhip_plan* plan, int column, const void* inter_dev, int dtype) {
    if (!plan || column < 0 || column >= plan->K) return fail(AFHIP_E_INVALID, "plan_bind_inter: no such column");
    if (plan->cols[(size_t)column].tf != TF_INTER) return fail(AFHIP_E_INVALID, "plan_bind_inter: column %d has no inter transform", column);
    if (!inter_dev || (dtype != AFHIP_F32 && dtype != AFHIP_F64)) return fail(AFHIP_E_INVALID, "plan_bind_inter: NULL array or bad dtype");
    int rcd = check_ptr_device(plan, inter_dev, "plan_bind_inter", "the second cube");
    if (rcd) return rcd;
    plan->cols[(size_t)column].inter = inter_dev;
    plan->cols[(size_t)column].inter_f32 = dtype == AFHIP_F32 ? 1 : 0;
    return AFHIP_OK;
}

extern "C" int64_t afhip_plan_workspace_bytes(const afhip_plan* plan) {
    if (!plan) return 0;
    return plan->ws_partial;
}

// rows of the sums buffer a spatial stage needs: the regions + the scratch rows of cut regions
static int64_t spmm_rows(const afhip_csr* csr) { return csr->R + csr->n_extra; }
static int64_t sums_bytes_of(const afhip_plan* plan, const afhip_csr* csr) {
    const int64_t Q = (int64_t)(plan->K + 1) * plan->desc.P;
    return (std::max<int64_t>(spmm_rows(csr) * Q, 1) * 8 + 255) / 256 * 256;
}

extern "C" int64_t afhip_plan_run_workspace_bytes(const afhip_plan* plan, const afhip_csr* csr) {
    if (!plan || !csr) return 0;
    return plan->ws_partial + plan->ws_panel + sums_bytes_of(plan, csr);
}

extern "C" int afhip_plan_describe(const afhip_plan* plan, char* buf, int buf_len) {
    if (!plan) return 0;
    int64_t min_len = INT64_MAX, max_len = 0;
    for (auto& c : plan->chunks) { min_len = std::min(min_len, c.k_hi - c.k_lo); max_len = std::max(max_len, c.k_hi - c.k_lo); }
    char tmp[1024], cg[48] = "";
    if (plan->last_counts_lanes > 0) snprintf(cg, sizeof cg, " last-run=count-gather/%d-lane", plan->last_counts_lanes);
    int n = snprintf(tmp, sizeof tmp,
                     "variant=%s pipe=%d vec=%d stat=%d slots=%d kmax=%d depth=%d | T=%lld cells=%lld K=%d G1=%lld P=%lld | "
                     "wg=%d tiles=%lld chunks=%zu (steps %lld..%lld) out_slots=%lld%s%s | workspace=%.1f MiB%s",
                     plan->variant->name, plan->variant->pipe, plan->variant->vec, plan->variant->stat, plan->variant->nthr,
                     plan->variant->kmax, plan->variant->depth, (long long)plan->desc.T, (long long)plan->desc.n_cells,
                     plan->K, (long long)plan->desc.G1, (long long)plan->desc.P, plan->wg, (long long)plan->tiles, plan->chunks.size(),
                     (long long)(plan->chunks.empty() ? 0 : min_len), (long long)max_len, (long long)plan->n_slots,
                     plan->packed ? (plan->pk.nw == 2 ? " packed-counts16" : " packed-counts32")
                                  : (plan->last_route == 1 ? " last-run=region-fused" : (plan->rf_plan_ok ? " region-fused-capable" : "")),
                     cg, (double)(plan->ws_partial + plan->ws_panel) / (1024.0 * 1024.0),
                     plan->last_ws == 1 ? " (caller-owned)" : (plan->last_ws == 2 ? " (plan-owned hipMalloc)" : ""));
    if (buf && buf_len > 0) snprintf(buf, buf_len, "%s", tmp);
    return n + 1;
}

// Layout of the run sums: slot-major [slots][runs][K + 1] (a period end's stores of one wave side by side: one or two cache lines per
// store instruction; k_rf_reduce then deals its (region, period) pairs period-major, so that neighbouring regions read neighbouring
// runs of ONE slot) or run-major [runs][slots][K + 1] (a run's periods side by side: k_rf_reduce reads whole lines — its time falls to
// a half ... a fifth — but every run a wave closes is then a cache line of its own at every period end: +0.25 ... 0.5 ms on the general
// streaming forms, next to nothing on the short-group forms).  Measured (profiles/r04_rf_layout.txt, both tables): run-major pays on the
// short-group forms from ~5e7 (run, period, column) gathers (daily sine_dd on 0.1 deg 12.3 -> 9.5 ms, weekly 5.12 -> 4.60, 6-hourly daily
// 3.21 -> 2.96) and on the general forms nowhere below 5e8 (float64 daily configs[1] panel 4.49 -> 4.87, float32 2.76 -> 2.90).
static bool rf_run_major(const afhip_plan* pl, const afhip_csr::RfTab* rf) {
    if (pl->rf_layout >= 0) return pl->rf_layout == 1;
    return (double)rf->n_runs * (double)pl->desc.P * (double)(pl->K + 1) >= (pl->variant->pair ? 5e7 : 5e8);
}

static int launch_temporal(afhip_plan* pl, const void* cube, double* partial, hipStream_t st, const afhip_csr::RfTab* rf = nullptr) {
    if (pl->chunks.empty()) return AFHIP_OK;
    for (int j = 0; j < pl->K; ++j)
        if (pl->cols[(size_t)j].tf == TF_INTER && !pl->cols[(size_t)j].inter)
            return fail(AFHIP_E_INVALID, "column %d multiplies by a second array (AFHIP_TF_INTER) that was never bound: call afhip_plan_bind_inter first", j);
    FusedArgs fa{};
    fa.cube = cube; fa.C = pl->desc.n_cells;
    fa.gtab = pl->d_gtab.p; fa.chunks = pl->d_chunks.p;
    fa.partial = partial; fa.K = pl->K; fa.nthr = pl->nthr;
    fa.n_tiles = (int32_t)pl->tiles;
    fa.xcd_remap = pl->xcd_remap;
    fa.sine_tab = nullptr;
    if (pl->has_sine) {
        int rc = sine_table_dev(pl->device, pl->variant->pair != 0 && pl->variant->quad == 0, &fa.sine_tab);
        if (rc) return rc;
    }
    for (int i = 0; i < pl->nthr; ++i) fa.thr[i] = pl->thr[(size_t)i];
    for (int i = pl->nthr; i < MAX_THR; ++i) {       // padded slots never fire
        fa.thr[i] = ThrSlot{};
        fa.thr[i].t0 = INFINITY; fa.thr[i].t1 = -INFINITY;
        fa.thr[i].t0f = INFINITY; fa.thr[i].t1f = -INFINITY;
    }
    for (int j = 0; j < pl->K; ++j) {
        const ColOp& c = pl->cols[(size_t)j];
        fa.cols[j] = c;
        fa.ccode[j] = (uint32_t)c.src | ((uint32_t)c.tf << 4) | ((uint32_t)(uint8_t)(int8_t)c.tf_iarg << 8);
    }
    fa.packed = pl->packed ? 1 : 0;
    fa.pk_nw = pl->pk.nw; fa.pk_mask = pl->pk.mask;
    for (int j = 0; j < MAX_COLS; ++j) { fa.pk_word[j] = pl->pk.word[j]; fa.pk_shift[j] = pl->pk.shift[j]; }
    dim3 grid((unsigned)pl->tiles, (unsigned)pl->chunks.size());
    void* args[] = {&fa};
    size_t lds = plan_lds_bytes(pl);
    const void* fn = pl->variant->fn;
    if (rf) {       // region-fused period ends: the twin variant, per-run sums into the partial area, a parking block per lane in LDS
        fn = pl->variant_rf->fn;
        lds = (lds + 15) / 16 * 16;
        fa.rf_lds_off = (int32_t)lds;
        lds += (size_t)pl->wg * (size_t)(pl->variant->vec * 16 + 16);      // RF_LANE_BYTES of afhip_kernels.h
        fa.rf_w = rf->w2.p; fa.rf_lane = rf->lane.p; fa.rf_tile = rf->tile.p; fa.rf_out = partial;
        fa.rf_slot_stride = rf_run_major(pl, rf) ? (int64_t)(pl->K + 1) : rf->n_runs * (pl->K + 1);
        fa.rf_run_stride = rf_run_major(pl, rf) ? pl->n_slots * (pl->K + 1) : (int64_t)(pl->K + 1);
        fa.rf_x = rf->n_xcells ? rf->xidx.p : nullptr; fa.rf_nx = rf->n_xcells;
        fa.rf_ex = partial + pl->n_slots * rf->n_runs * (pl->K + 1);           // behind the run sums
    }
    if (pl->variant->hb) {
        fa.hb_n = pl->hb_n; fa.hb_c1 = pl->hb_c1; fa.hb_c0 = pl->hb_c0;
        fa.hb_c1f = (float)pl->hb_c1; fa.hb_c0f = (float)pl->hb_c0;
        fa.hb_shift = pl->wg == 64 ? 6 : (pl->wg == 128 ? 7 : 8);
        for (int i = 0; i < MAX_THR; ++i) fa.hb_bin_of_slot[i] = pl->hb_bin_of_slot[i];
        for (int k = 0; k <= pl->hb_n; ++k) {
            const double t = pl->hb_edge[k];
            const float f = (float)t;
            fa.hb_edge[k] = t;
            fa.hb_dn[k] = (double)f > t ? std::nextafterf(f, -INFINITY) : f;     // largest float <= t
            fa.hb_up[k] = (double)f < t ? std::nextafterf(f, INFINITY) : f;      // smallest float >= t
        }
        fa.hb_w = pl->hb_w; fa.hb_lo0 = pl->hb_lo0; fa.hb_gl = pl->hb_gl; fa.hb_gh = pl->hb_gh;
        fa.hb_wf = (float)pl->hb_w; fa.hb_lo0f = (float)pl->hb_lo0; fa.hb_glf = (float)pl->hb_gl; fa.hb_ghf = (float)pl->hb_gh;
        fa.hb_c0b = pl->hb_c0b; fa.hb_c0bf = (float)pl->hb_c0b;
    }
    HIP_TRY(hipLaunchKernel(fn, grid, dim3((unsigned)pl->wg), args, lds, st));
    return AFHIP_OK;
}

static int launch_combine(afhip_plan* pl, const double* partial, double* cells, double* panel, hipStream_t st) {
    const int64_t C = pl->desc.n_cells, P = pl->desc.P;
    if (P == 0) return AFHIP_OK;
    CombineArgs ca{};
    ca.partial = partial; ca.slot_ptr = pl->d_slot_ptr.p; ca.outer_bounds = pl->d_ob.p;
    ca.cells_out = cells; ca.panel = panel; ca.C = C; ca.P = P; ca.K = pl->K;
    ca.pk = pl->packed ? pl->pk : PackFmt{};
    for (int j = 0; j < pl->K; ++j) {
        ca.outer[j] = pl->cols[(size_t)j].outer;
        ca.round_final[j] = (pl->cols[(size_t)j].rounding & AFHIP_ROUND_FINAL) ? 1 : 0;
    }
    if (P >= 4 && panel) {      // many periods: the tiled kernel writes the panel in contiguous runs
        dim3 grid((unsigned)((C + CT_CELLS - 1) / CT_CELLS), (unsigned)((P + CT_PER - 1) / CT_PER));
        if (grid.y > 65535) return fail(AFHIP_E_UNSUPPORTED, "more than 524,280 output periods in one call");
        hipLaunchKernelGGL(k_combine_slots_tiled, grid, dim3(WG), 0, st, ca);
    } else {
        dim3 grid((unsigned)((C + WG - 1) / WG), (unsigned)std::min<int64_t>(P, 65535));   // strides over periods
        hipLaunchKernelGGL(k_combine_slots, grid, dim3(WG), 0, st, ca);
    }
    HIP_TRY(hipGetLastError());
    return AFHIP_OK;
}

// sums[row][p][K + 1] straight from partial (k_csr_spmm_slots): slot merge, shared validity and the weighted sums of every
// (segment, period) in one pass; then the pieces of cut rows.  The lanes per (segment, period) follow the table's mean
// segment length — 64 for county-sized rows on a fine grid (and then bit-identical to combine + k_csr_spmm_wave), 8 for
// tables whose regions hold a handful of cells.
static int launch_spmm_slots(afhip_plan* pl, const afhip_csr* csr, const double* partial, hipStream_t st,
                             double* num_dev, double* den_dev, double* res_dev, bool* divided) {
    const int64_t P = pl->desc.P, K = pl->K, Q = (K + 1) * P;
    if (csr->nseg * P == 0) return AFHIP_OK;
    SlotSpmmArgs sa{};
    sa.seg_ptr = csr->seg_ptr.p; sa.dst = csr->seg_dst.p; sa.cols = csr->cols.p; sa.w = csr->w.p;
    sa.partial = partial; sa.slot_ptr = pl->d_slot_ptr.p; sa.outer_bounds = pl->d_ob.p; sa.out = pl->sums;
    sa.nseg = csr->nseg; sa.P = P; sa.C = pl->desc.n_cells; sa.K = (int32_t)K;
    sa.p_major = pl->slot_spmm_order >= 0 ? pl->slot_spmm_order : (P >= 8 ? 1 : 0);
    // no row of the table is cut: a segment IS a region, and the lane that holds its K + 1 sums finishes the panel (no divide kernel)
    *divided = csr->n_split == 0 && csr->nseg == csr->R && !pl->no_slots_divide;
    if (*divided) { sa.num = num_dev; sa.den = den_dev; sa.res = res_dev; sa.R = csr->R; }
    for (int j = 0; j < pl->K; ++j) {
        sa.outer[j] = pl->cols[(size_t)j].outer;
        sa.round_final[j] = (pl->cols[(size_t)j].rounding & AFHIP_ROUND_FINAL) ? 1 : 0;
    }
    const int64_t mean_len = csr->nnz / std::max<int64_t>(csr->nseg, 1);
    int sub = mean_len > 32 ? 64 : (mean_len > 16 ? 32 : (mean_len > 8 ? 16 : 8));
    if (pl->slot_spmm_sub) sub = pl->slot_spmm_sub;
    const int64_t pairs = csr->nseg * P;
    const int64_t blocks = (pairs * sub + WG - 1) / WG;
    if (blocks > INT32_MAX) return fail(AFHIP_E_UNSUPPORTED, "plan_run: %lld (segment, period) pairs in one call", (long long)pairs);
#define AFHIP_SLOTS_SUB(KB)                                                                                                   \
    switch (sub) {                                                                                                            \
        case 8: hipLaunchKernelGGL((k_csr_spmm_slots<KB, 8>), dim3((unsigned)blocks), dim3(WG), 0, st, sa); break;            \
        case 16: hipLaunchKernelGGL((k_csr_spmm_slots<KB, 16>), dim3((unsigned)blocks), dim3(WG), 0, st, sa); break;          \
        case 32: hipLaunchKernelGGL((k_csr_spmm_slots<KB, 32>), dim3((unsigned)blocks), dim3(WG), 0, st, sa); break;          \
        default: hipLaunchKernelGGL((k_csr_spmm_slots<KB, 64>), dim3((unsigned)blocks), dim3(WG), 0, st, sa); break;          \
    }
    if (K <= 2) { AFHIP_SLOTS_SUB(2) }
    else if (K <= 4) { AFHIP_SLOTS_SUB(4) }
    else if (K <= 8) { AFHIP_SLOTS_SUB(8) }
    else { AFHIP_SLOTS_SUB(16) }
#undef AFHIP_SLOTS_SUB
    HIP_TRY(hipGetLastError());
    if (csr->n_split) {
        const int64_t n = csr->n_split * Q;
        hipLaunchKernelGGL(k_csr_combine_segments, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, pl->sums, csr->split_row.p,
                           csr->split_ptr.p, csr->R, Q, csr->n_split);
        HIP_TRY(hipGetLastError());
    }
    return AFHIP_OK;
}

// The scratch of a run: the caller's block when given (checked for size, alignment and device: nothing below allocates, frees
// or synchronises then), else the plan's own, grown at need — the outgrown block is kept until afhip_plan_destroy, because
// kernels of an earlier run on another stream may still read it and a hipFree would wait for the whole device.
static int ensure_ws(afhip_plan* pl, int64_t bytes, void* user_ws, int64_t user_bytes, const char* who, char** base) {
    if (user_ws) {
        if (user_bytes < bytes)
            return fail(AFHIP_E_INVALID, "%s: the workspace holds %lld bytes, the run needs %lld (afhip_plan_run_workspace_bytes / afhip_plan_workspace_bytes)",
                        who, (long long)user_bytes, (long long)bytes);
        if ((uintptr_t)user_ws % 256 != 0) return fail(AFHIP_E_INVALID, "%s: the workspace must be 256-byte aligned", who);
        int rc = check_ptr_device(pl, user_ws, who, "the workspace");
        if (rc) return rc;
        *base = (char*)user_ws;
        pl->last_ws = 1;
        return AFHIP_OK;
    }
    if (pl->own_ws_bytes < bytes) {
        if (pl->own_ws) { pl->retired.push_back(pl->own_ws); pl->own_ws = nullptr; pl->own_ws_bytes = 0; }
        HIP_TRY(hipMalloc(&pl->own_ws, (size_t)bytes));
        pl->own_ws_bytes = bytes;
    }
    *base = (char*)pl->own_ws;
    pl->last_ws = 2;
    return AFHIP_OK;
}

extern "C" int afhip_plan_run_temporal(afhip_plan* plan, const void* cube_dev, double* cells_dev,
                                       void* workspace_dev, int64_t workspace_bytes, void* stream) {
    if (!plan || !cube_dev || !cells_dev) return fail(AFHIP_E_INVALID, "plan_run_temporal: NULL argument");
    int rcd = check_cube_device(plan, cube_dev, "plan_run_temporal");
    if (!rcd) rcd = check_ptr_device(plan, cells_dev, "plan_run_temporal", "the per-cell output");
    if (rcd) return rcd;
    GUARD_DEVICE(plan->device);
    hipStream_t st = (hipStream_t)stream;
    char* base;
    int rc = ensure_ws(plan, plan->ws_partial, workspace_dev, workspace_bytes, "plan_run_temporal", &base);
    if (rc) return rc;
    double* partial = (double*)base;
    if ((rc = launch_temporal(plan, cube_dev, partial, st))) return rc;
    return launch_combine(plan, partial, cells_dev, nullptr, st);
}

extern "C" int afhip_plan_run(afhip_plan* plan, const void* cube_dev, const afhip_csr* csr,
                              double* num_dev, double* den_dev, double* res_dev, double* cells_dev,
                              void* workspace_dev, int64_t workspace_bytes, void* stream, float* kernel_ms) {
    if (!plan || !cube_dev || !csr || !res_dev) return fail(AFHIP_E_INVALID, "plan_run: NULL argument");
    if (csr->n_cells != plan->desc.n_cells)
        return fail(AFHIP_E_INVALID, "plan_run: CSR has %lld cells, plan has %lld", (long long)csr->n_cells, (long long)plan->desc.n_cells);
    if (csr->device != plan->device)
        return fail(AFHIP_E_INVALID, "plan_run: the CSR lives on device %d, the plan on device %d", csr->device, plan->device);
    int rcd = check_cube_device(plan, cube_dev, "plan_run");
    if (!rcd) rcd = check_ptr_device(plan, res_dev, "plan_run", "the result panel");
    if (!rcd && num_dev) rcd = check_ptr_device(plan, num_dev, "plan_run", "the numerator panel");
    if (!rcd && den_dev) rcd = check_ptr_device(plan, den_dev, "plan_run", "the denominator panel");
    if (!rcd && cells_dev) rcd = check_ptr_device(plan, cells_dev, "plan_run", "the per-cell output");
    if (rcd) return rcd;
    GUARD_DEVICE(plan->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t K = plan->K, P = plan->desc.P, Q = (K + 1) * P;
    // partial | panel | sums[rows][Q] in one block: the caller's workspace when given, else the plan's own
    char* base;
    int rc = ensure_ws(plan, afhip_plan_run_workspace_bytes(plan, csr), workspace_dev, workspace_bytes, "plan_run", &base);
    if (rc) return rc;
    double* partial = (double*)base;
    double* panel = (double*)(base + plan->ws_partial);
    plan->sums = (double*)(base + plan->ws_partial + plan->ws_panel);
    if (kernel_ms) {
        for (auto& e : plan->ev) if (!e) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventRecord(plan->ev[0], st));
    }
    const bool exact = plan->desc.exact_order != 0 || spmm_serial_env();
    // region-fused period ends: the plan allows it, no per-cell output is wanted, and the table's runs exist (built on first use) and
    // fit the partial area (n_runs * 3 <= cells and K + 1 <= 2 K ... checked in bytes)
    const afhip_csr::RfTab* rf = nullptr;
    if (plan->rf_plan_ok && !exact && !cells_dev && !plan->packed) {
        rf = rf_table(const_cast<afhip_csr*>(csr), plan->variant->vec);
        if (rf && plan->n_slots * (rf->n_runs + rf->n_xcells) * (K + 1) * 8 > plan->ws_partial) rf = nullptr;
    }
    const bool prof = !plan->prof_ev.empty() && (size_t)(2 * plan->prof_count + 1) < plan->prof_ev.size();
    if (prof) HIP_TRY(hipEventRecord(plan->prof_ev[(size_t)(2 * plan->prof_count)], st));
    if ((rc = launch_temporal(plan, cube_dev, partial, st, rf))) return rc;
    if (prof) { HIP_TRY(hipEventRecord(plan->prof_ev[(size_t)(2 * plan->prof_count + 1)], st)); ++plan->prof_count; }
    if (kernel_ms) HIP_TRY(hipEventRecord(plan->ev[1], st));
    plan->last_route = rf ? 1 : 0;
    plan->last_counts_lanes = -1;
    bool divided = false;           // the count gather wrote num / den / res itself
    if (rf) {
        // a region's runs added in run order -> sums[r][p][K + 1] (no pieces: rows [0, R) only)
        const int64_t n = csr->R * P * (K + 1);
        if (n) {
            uint32_t mean_mask = 0;
            for (int j = 0; j < plan->K; ++j) if (plan->cols[(size_t)j].outer == OUT_MEAN) mean_mask |= 1u << j;
            hipLaunchKernelGGL(k_rf_reduce, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, (const double*)partial, rf->reg_ptr.p, rf->reg_runs.p,
                               plan->d_slot_ptr.p, plan->d_ob.p, mean_mask, (const double*)(partial + plan->n_slots * rf->n_runs * (K + 1)), rf->n_xcells,
                               rf->xreg_ptr.p, rf->xcell.p, rf->xw.p, plan->sums, csr->R, P, (int)(K + 1), rf->n_runs,
                               rf_run_major(plan, rf) ? (int64_t)(K + 1) : rf->n_runs * (K + 1), rf_run_major(plan, rf) ? plan->n_slots * (K + 1) : (int64_t)(K + 1),
                               plan->rf_reduce_order >= 0 ? plan->rf_reduce_order : ((!rf_run_major(plan, rf) && P >= 8) ? 1 : 0));
            HIP_TRY(hipGetLastError());
        }
    } else if (plan->packed && !cells_dev && plan->n_slots <= P && plan->counts_spmm) {
        // bin-count plan, no per-cell output wanted: the weighted sums gather the packed counts directly
        // (every period has at most one slot: single-level plans are never split); long rows in segments unless exact
        const int64_t nv = exact ? csr->R : csr->nseg, nq = nv * P;
        if (nq) {
            // rows that are never cut (exact order, or a table without long rows): the gather finishes the panel itself, no divide kernel
            divided = (exact || csr->n_split == 0) && nv == csr->R && !plan->no_counts_divide;
            // lanes per pair: one (the table's order: `exact_order`), or a group of lanes over the row's entries with the pairs dealt period-major
            // (one lane per pair leaves a job of few periods with few threads, each walking its row alone: annual bins on 3,100 county-sized
            // regions 0.365 ms against 0.030 with sixteen lanes per pair; many periods turn it around — the group's lanes idle on short
            // rows and every pair pays the group's adds: configs[3], 17 entries x 251 periods, 0.217 against 0.43.  Measured crossover:
            // about as many periods as a row has entries; profiles/r04_counts_gather.txt)
            const int64_t nnz_rows = csr->h_indptr.empty() ? 0 : csr->h_indptr.back();
            const double entries = nv > 0 ? (double)nnz_rows / (double)nv : 0.0;
            const int sub_rule = ((double)P < entries) ? (entries < 12.0 ? 8 : 16) : 0;
            const int sub = exact ? 0 : (plan->counts_spmm_sub >= 0 ? plan->counts_spmm_sub : sub_rule);
            plan->last_counts_lanes = (sub == 4 || sub == 8 || sub == 16) ? sub : 1;
            const int64_t* ip = exact ? csr->indptr.p : csr->seg_ptr.p;
            const int32_t* dr = exact ? (const int32_t*)nullptr : csr->seg_dst.p;
            double* o_num = divided ? num_dev : (double*)nullptr; double* o_den = divided ? den_dev : (double*)nullptr; double* o_res = divided ? res_dev : (double*)nullptr;
            if (sub == 4 || sub == 8 || sub == 16) {
                const dim3 gr((unsigned)((nq * sub + WG - 1) / WG)), bl(WG);
                if (sub == 4) hipLaunchKernelGGL((k_csr_spmm_counts_sub<4>), gr, bl, 0, st, ip, dr, csr->cols.p, csr->w.p, (const void*)partial, plan->d_slot_ptr.p, plan->sums, nv, P, (int)K, plan->desc.n_cells, plan->pk, o_num, o_den, o_res);
                else if (sub == 8) hipLaunchKernelGGL((k_csr_spmm_counts_sub<8>), gr, bl, 0, st, ip, dr, csr->cols.p, csr->w.p, (const void*)partial, plan->d_slot_ptr.p, plan->sums, nv, P, (int)K, plan->desc.n_cells, plan->pk, o_num, o_den, o_res);
                else hipLaunchKernelGGL((k_csr_spmm_counts_sub<16>), gr, bl, 0, st, ip, dr, csr->cols.p, csr->w.p, (const void*)partial, plan->d_slot_ptr.p, plan->sums, nv, P, (int)K, plan->desc.n_cells, plan->pk, o_num, o_den, o_res);
            } else
            hipLaunchKernelGGL(k_csr_spmm_counts, dim3((unsigned)((nq + WG - 1) / WG)), dim3(WG), 0, st, ip, dr, csr->cols.p,
                               csr->w.p, (const void*)partial, plan->d_slot_ptr.p, plan->sums, nv, P, (int)K, plan->desc.n_cells, plan->pk,
                               o_num, o_den, o_res);
            HIP_TRY(hipGetLastError());
            if (!exact && csr->n_split) {
                const int64_t n = csr->n_split * Q;
                hipLaunchKernelGGL(k_csr_combine_segments, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, plan->sums, csr->split_row.p,
                                   csr->split_ptr.p, csr->R, Q, csr->n_split);
                HIP_TRY(hipGetLastError());
            }
        }
    } else if (!exact && !cells_dev && !plan->packed && K > 0 && !plan->no_slot_spmm) {
        // no per-cell output wanted, no table-order promise: the weighted sums gather the slots directly (no panel)
        if ((rc = launch_spmm_slots(plan, csr, partial, st, num_dev, den_dev, res_dev, &divided))) return rc;
    } else {
        if ((rc = launch_combine(plan, partial, cells_dev, panel, st))) return rc;
        if ((rc = launch_spmm(csr, panel, plan->sums, Q, st, exact))) return rc;
    }
    const int64_t n = K > 0 ? csr->R * P : 0;                 // one thread per (region, period)
    if (n && !divided) {
        hipLaunchKernelGGL(k_panel_divide, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, plan->sums, num_dev,
                           den_dev, res_dev, csr->R, P, (int)K);
        HIP_TRY(hipGetLastError());
    }
    if (kernel_ms) {
        HIP_TRY(hipEventRecord(plan->ev[2], st));
        HIP_TRY(hipEventSynchronize(plan->ev[2]));
        HIP_TRY(hipEventElapsedTime(&kernel_ms[0], plan->ev[0], plan->ev[1]));
        HIP_TRY(hipEventElapsedTime(&kernel_ms[1], plan->ev[0], plan->ev[2]));
    }
    return AFHIP_OK;
}

extern "C" int afhip_plan_profile_begin(afhip_plan* plan, int64_t max_launches) {
    if (!plan || max_launches < 0) return fail(AFHIP_E_INVALID, "plan_profile_begin: bad arguments");
    GUARD_DEVICE(plan->device);
    for (auto& e : plan->prof_ev) if (e) (void)hipEventDestroy(e);
    plan->prof_ev.assign((size_t)(2 * max_launches), nullptr);
    for (auto& e : plan->prof_ev) HIP_TRY(hipEventCreate(&e));
    plan->prof_count = 0;
    return AFHIP_OK;
}

extern "C" int64_t afhip_plan_profile_end(afhip_plan* plan, float* ms_out, int64_t cap) {
    if (!plan) return 0;
    DeviceGuard guard__(plan->device);
    const int64_t n = plan->prof_count;
    for (int64_t i = 0; i < n && i < cap; ++i) {
        if (hipEventSynchronize(plan->prof_ev[(size_t)(2 * i + 1)]) != hipSuccess ||
            hipEventElapsedTime(&ms_out[i], plan->prof_ev[(size_t)(2 * i)], plan->prof_ev[(size_t)(2 * i + 1)]) != hipSuccess) {
            fail(AFHIP_E_HIP, "plan_profile_end: event query failed");
            return -1;
        }
    }
    for (auto& e : plan->prof_ev) if (e) (void)hipEventDestroy(e);
    plan->prof_ev.clear();
    plan->prof_count = 0;
    return n;
}

// ---------------------------------------------------------------------------------------
// standalone grouped reducers (drop-in for the numba kernels)
// ---------------------------------------------------------------------------------------
static int run_group(const void* cube_dev, int dtype, int64_t T, int64_t n_cells, const int64_t* bounds,
                     int64_t G, int code, const double* ddargs, int64_t D, void* out_dev, void* stream) {
    if (!cube_dev || !bounds || !out_dev) return fail(AFHIP_E_INVALID, "group kernel: NULL argument");
    if (G < 0 || D <= 0) return fail(AFHIP_E_INVALID, "group kernel: bad G/D");
    if (G == 0) return AFHIP_OK;
    GUARD_DEVICE(pointer_device(cube_dev));          // the temporary plans are created, run and freed on the cube's device
    hipStream_t st = (hipStream_t)stream;
    std::vector<int64_t> ob((size_t)G + 1);
    for (int64_t g = 0; g <= G; ++g) ob[(size_t)g] = g;
    // D can exceed one pass's slot / column budget (the reference loops over any number of ddargs rows, nb_kernels.py:166,190,215):
    // passes of <= per_pass columns, each writing its own columns [d0, d0 + n) of out[G][cell][D]
    const int64_t per_pass = std::min<int64_t>(MAX_COLS, MAX_THR);
    for (int64_t d0 = 0; d0 < D; d0 += per_pass) {
        const int64_t n = std::min(per_pass, D - d0);
        std::vector<afhip_column> cols((size_t)n);
        for (int64_t d = 0; d < n; ++d) {
            afhip_column c{};
            c.inner = code; c.transform = AFHIP_TF_NONE; c.outer = AFHIP_IDENTITY;
            if (ddargs) { c.inner_args[0] = ddargs[(d0 + d) * 3]; c.inner_args[1] = ddargs[(d0 + d) * 3 + 1]; c.inner_args[2] = ddargs[(d0 + d) * 3 + 2]; }
            cols[(size_t)d] = c;
        }
        afhip_plan_desc desc{};
        desc.T = T; desc.n_cells = n_cells; desc.dtype = dtype; desc.K = (int32_t)n; desc.G1 = G;
        desc.inner_bounds = bounds; desc.P = G; desc.outer_bounds = ob.data(); desc.columns = cols.data();
        afhip_plan* pl = nullptr;
        int rc = afhip_plan_create(&desc, &pl);
        if (rc) return rc;
        char* base;
        if ((rc = ensure_ws(pl, pl->ws_partial, nullptr, 0, "group kernel", &base))) { delete pl; return rc; }
        double* partial = (double*)base;
        if ((rc = launch_temporal(pl, cube_dev, partial, st))) { delete pl; return rc; }
        dim3 grid((unsigned)((n_cells + WG - 1) / WG), (unsigned)std::min<int64_t>(G, 65535));
        if (dtype == AFHIP_F32)
            hipLaunchKernelGGL(k_slots_to_block<float>, grid, dim3(WG), 0, st, partial, pl->d_slot_ptr.p, (float*)out_dev, n_cells, G, (int)n, (int)D, (int)d0, pl->packed ? pl->pk : PackFmt{});
        else
            hipLaunchKernelGGL(k_slots_to_block<double>, grid, dim3(WG), 0, st, partial, pl->d_slot_ptr.p, (double*)out_dev, n_cells, G, (int)n, (int)D, (int)d0, pl->packed ? pl->pk : PackFmt{});
        hipError_t e = hipGetLastError();
        // the plan owns the scratch the kernels are still reading: drain before freeing it
        hipError_t e2 = hipStreamSynchronize(st);
        delete pl;
        if (e != hipSuccess) return fail(AFHIP_E_HIP, "k_slots_to_block launch failed: %s", hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(AFHIP_E_HIP, "stream synchronize failed: %s", hipGetErrorString(e2));
    }
    return AFHIP_OK;
}

extern "C" int afhip_group_stat(const void* cube_dev, int dtype, int64_t T, int64_t n_cells,
                                const int64_t* bounds, int64_t G, int code, void* out_dev, void* stream) {
    if (code < AFHIP_MEAN || code > AFHIP_NANMEAN) return fail(AFHIP_E_INVALID, "group_stat: code %d is not a stat reducer", code);
    return run_group(cube_dev, dtype, T, n_cells, bounds, G, code, nullptr, 1, out_dev, stream);
}
extern "C" int afhip_group_dd(const void* cube_dev, int dtype, int64_t T, int64_t n_cells, const int64_t* bounds,
                              int64_t G, const double* ddargs, int64_t D, void* out_dev, void* stream) {
    if (!ddargs) return fail(AFHIP_E_INVALID, "group_dd: ddargs is NULL");
    return run_group(cube_dev, dtype, T, n_cells, bounds, G, AFHIP_DD, ddargs, D, out_dev, stream);
}
extern "C" int afhip_group_bins(const void* cube_dev, int dtype, int64_t T, int64_t n_cells, const int64_t* bounds,
                                int64_t G, const double* ddargs, int64_t D, void* out_dev, void* stream) {
    if (!ddargs) return fail(AFHIP_E_INVALID, "group_bins: ddargs is NULL");
    return run_group(cube_dev, dtype, T, n_cells, bounds, G, AFHIP_BINS, ddargs, D, out_dev, stream);
}
extern "C" int afhip_group_sine_dd(const void* cube_dev, int dtype, int64_t T, int64_t n_cells, const int64_t* bounds,
                                   int64_t G, const double* ddargs, int64_t D, void* out_dev, void* stream) {
    if (!ddargs) return fail(AFHIP_E_INVALID, "group_sine_dd: ddargs is NULL");
    return run_group(cube_dev, dtype, T, n_cells, bounds, G, AFHIP_SINE_DD, ddargs, D, out_dev, stream);
}
