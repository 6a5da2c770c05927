// afhip_variants.h — the menu of compiled k_fused_temporal specialisations.
//
// The hot per-element loop must be straight-line code with every accumulator in a
// register, so the accumulator counts are template parameters.  gen_variants.py writes
// one translation unit per group of instantiations (compiled in parallel) plus the table
// below; find_variant() picks the cheapest instantiation that covers a lowered plan.
#pragma once
#include <stdint.h>

namespace afhip {

struct Variant {
    int dtype;    // AFHIP_F32 / AFHIP_F64
    int pipe;     // 0 direct loads, 1 LDS-DMA ring
    int vec;      // cells per lane
    int stat;     // 0 none, 1 sum, 2 sum+min+max, 3 NaN-skipping sum+count+min+max
    int nthr;     // threshold slots
    int kmax;     // columns
    int depth;    // LDS ring depth in rows (pipe 1)
    int nt;       // 1: non-temporal cache policy on the streaming loads
    int production;   // 1: part of the default menu; 0: tuning arm only
    int tki;          // 1: integer bin counters (all threshold slots must be bins)
    int sl;           // 1: single-level plans only (no outer accumulators)
    int hb;           // 1: LDS-histogram bins (contiguous equal-width partition)
    int ha;           // 1: ... whose edges are exactly representable: computed, not read from the LDS table
    int pair;         // 1: plans whose inner groups all hold exactly two rows ((tmin, tmax) pairs)
    int ss;           // ... with the lean group end: 1 = columns mean | sum | min | max | sine_dd -> (integer power) -> sum | mean;
                      //     2 = every column a plain sine_dd -> sum | mean (the tightest form)
    int quad;         // 1: ... for inner groups of exactly FOUR rows (6-hourly data) instead of two, 2: of exactly THREE rows (8-hourly), 3: of one to four rows, mixed; lean form only
    int rf;           // 1: region-fused period ends compiled in (the twin of the variant with the same other fields)
    const void* fn;
    const char* name;
};

const Variant* variants_table(int* n);   // generated (variants_table.hip)
const char* variants_menu();             // "full" (the production menu), "arms" (+ the tuning arms) or "dev"

// tuning: 0 = the default choice below; otherwise an explicit arm
//         pipe*1000 + vec*100 + depth  (+10000: default cache policy instead of nt)
//         e.g. 1404 LDS ring, 4 cells per lane, depth 4;  108 direct loads, 1 cell per lane, 8 rows in flight
inline const Variant* find_variant(int dtype, int pipe, int stat, int nthr, int K, int tuning, int vec = 0,
                                   bool all_bins = false, bool single_level = false, bool partition = false, bool arith = false,
                                   bool pairs = false, int lean = 0, int depth_hint = 0, int quads = 0, bool rf = false) {
    const Variant* best = nullptr;
    long best_cost = 0;
    int n = 0;
    const Variant* tab = variants_table(&n);
    int want_nt = 1, want_pipe = pipe, want_vec = -1, want_depth = -1;
    if (tuning > 0) {
        int t = tuning;
        if (t >= 10000) { want_nt = 0; t -= 10000; }
        want_pipe = t / 1000; want_vec = (t % 1000) / 100; want_depth = t % 100;
    }
    for (int i = 0; i < n; ++i) {
        const Variant& v = tab[i];
        if (v.dtype != dtype || v.stat < stat || v.nthr < nthr || v.kmax < K || (v.rf != 0) != rf) continue;
        if ((v.tki && !(all_bins && nthr > 0)) || (v.sl && !single_level) || (v.hb && !partition) || (v.ha && !arith)) continue;
        if (v.pair && !((pairs || quads) && nthr == 0)) continue;
        if ((v.pair && v.quad != quads) || (quads && !v.pair)) continue;    // a three- / four-row plan takes the variants of its group length only, and vice versa
        if (v.ss > lean) continue;                      // a lean variant needs a plan that qualifies for its form (2 implies 1)
        if (v.pipe != want_pipe || v.nt != want_nt) continue;
        if (tuning > 0) {
            if (v.vec != want_vec || v.depth != want_depth) continue;
        } else if (!v.production || (vec > 0 && v.vec != vec)) continue;
        // specialised forms (integer bins, single level) are cheaper than the general one
        const long cost = (long)v.nthr * 1000 + (long)v.kmax * 10 + v.stat - (v.tki ? 400 : 0) - (v.sl ? 5 : 0) - (v.hb ? 300 : 0) - (v.ha ? 50 : 0) - (v.pair ? 5 : 0) - 3 * v.ss
                          + ((depth_hint > 0 && v.depth != depth_hint) ? 1 : 0);      // among equals, the burst depth that measured best for the shape
        if (!best || cost < best_cost) { best = &v; best_cost = cost; }
    }
    return best;
}

}  // namespace afhip
