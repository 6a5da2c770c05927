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
    const void* fn;
    const char* name;
};

const Variant* variants_table(int* n);   // generated (variants_table.hip)

// tuning: 0 default | 1 force direct scalar loads | 2 direct vector loads (16 B per lane)
//         | 4 (default), 8, 16 LDS ring of that depth
inline const Variant* find_variant(int dtype, int pipe, int stat, int nthr, int K, int tuning) {
    const Variant* best = nullptr;
    long best_cost = 0;
    const int want_vec16 = (dtype == 0) ? 4 : 2;
    int g_n_variants = 0;
    const Variant* g_variants = variants_table(&g_n_variants);
    for (int i = 0; i < g_n_variants; ++i) {
        const Variant& v = g_variants[i];
        if (v.dtype != dtype || v.stat < stat || v.nthr < nthr || v.kmax < K) continue;
        if (tuning == 1) { if (v.pipe != 0 || v.vec != 1) continue; }
        else if (tuning == 2) { if (v.pipe != 0 || v.vec != want_vec16) continue; }
        else if (tuning >= 4) { if (v.pipe != 1 || v.depth != tuning) continue; }
        else {
            if (v.pipe != pipe) continue;
            if (pipe == 0 && v.vec != 1) continue;
            if (pipe == 1 && v.depth != 4) continue;
        }
        const long cost = (long)v.nthr * 1000 + (long)v.kmax * 10 + v.stat;
        if (!best || cost < best_cost) { best = &v; best_cost = cost; }
    }
    return best;
}

}  // namespace afhip
