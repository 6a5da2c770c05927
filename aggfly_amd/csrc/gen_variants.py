#!/usr/bin/env python3
"""Writes the translation units that instantiate k_fused_temporal (afhip_kernels.h).

Usage: gen_variants.py OUTDIR [--menu full|arms|dev] [--per-file N]      (full: the production menu; arms: + the tuning arms)

Each generated ``variants_NN.hip`` holds a handful of explicit instantiations so that
``make -j`` compiles them in parallel; ``variants_table.hip`` collects the table that
``find_variant`` (afhip_variants.h) searches.
"""
import os
import sys

CT = {0: "float", 1: "double"}


def menu(kind):
    """(dtype, pipe, vec, stat, nthr, kmax, depth, feat, production)"""
    out = []
    vec16 = {0: 4, 1: 2}

    def add(dtype, pipe, vec, stat, nthr, kmax, depth, nt=1, prod=1, tki=0, sl=0, hb=0, ha=0, pair=0, ss=0, quad=0, rf=0, tri=0, rag=0):
        # sine degree days ride on the min/max accumulators; generic pow() only in the
        # all-purpose (STAT 3) variants; bit 2 = nt cache policy on the streaming loads;
        # bit 3 = integer bin counters; bit 4 = single-level plan (no outer accumulators)
        feat = {0: 0, 1: 0, 2: 1, 3: 3}[stat] | (4 if nt else 0) | (8 if tki else 0) | (16 if sl else 0) | (32 if hb else 0) | (64 if ha else 0) | (128 if pair else 0) | (256 if ss else 0) | (512 if ss == 2 else 0) | (1024 if quad else 0) | (2048 if rf else 0) | (4096 if tri else 0) | (8192 if rag else 0)
        key = (dtype, pipe, vec, stat, nthr, kmax, depth, feat)
        for i, v in enumerate(out):
            if v[:8] == key:
                out[i] = key + (max(v[8], prod),)
                return
        out.append(key + (prod,))

    shapes_all = [(stat, nthr, kmax) for stat in (0, 1, 3) for nthr in (0, 1, 4, 16) for kmax in (2, 6, 16)
                  if not (stat == 0 and nthr == 0)] + [(2, 0, 2), (2, 0, 6)]
    headline = ((1, 1, 6), (1, 0, 2), (2, 0, 2), (0, 16, 16))
    for dtype in (0, 1):
        if kind == "dev":
            add(dtype, 0, 1, 3, 16, 16, 4)
            add(dtype, 0, 1, 1, 1, 6, 4)
            add(dtype, 1, vec16[dtype], 1, 1, 6, 4)
            add(dtype, 1, vec16[dtype], 3, 16, 16, 4)
            continue
        for (stat, nthr, kmax) in shapes_all:
            # production menu, chosen by measurement on MI355X (profiles/r01_sweep_load_arms.txt):
            #   f64: direct 8-byte nt loads, one cell per lane, 4 rows in flight (6.5 TB/s on
            #        configs[1]); the LDS-DMA ring with 2 cells per lane is kept for short inner groups (and as an arm)
            #   f32: direct 8-byte nt loads, two cells per lane, 8 rows in flight; odd row
            #        lengths fall back to one cell per lane
            if dtype == 1:
                add(dtype, 0, 1, stat, nthr, kmax, 4)
                add(dtype, 1, 2, stat, nthr, kmax, 4)
            else:
                add(dtype, 0, 2, stat, nthr, kmax, 8)
                add(dtype, 0, 1, stat, nthr, kmax, 8)
                # short inner groups (2-4 steps: tmin/tmax pairs, 6-hourly data): the LDS-DMA ring prefetches
                # across group ends, the direct path cannot (3.5 vs 4.3 TB/s at 2 steps, 4.4 vs 5.6 at 4)
                add(dtype, 1, 4, stat, nthr, kmax, 4)
        # bins-heavy and multi-threshold single-level plans (CMIP6 temperature bins, configs[3])
        for nthr in (4, 16):
            for kmax in (6, 16):
                if kmax < nthr and nthr == 16:
                    continue
                for (pipe, vec, depth) in (((0, 1, 4), (1, 2, 4)) if dtype == 1 else ((0, 2, 8), (0, 1, 8))):
                    add(dtype, pipe, vec, 0, nthr, kmax, depth, tki=1, sl=1)
                    add(dtype, pipe, vec, 0, nthr, kmax, depth, tki=0, sl=1)
                    add(dtype, pipe, vec, 1, nthr, kmax, depth, tki=1, sl=0)
        # (tmin, tmax) pairs per day (configs[4]): every inner group is two rows
        for kmax in (2, 6):
            for (vec, depth) in (((1, 4), (1, 8)) if dtype == 1 else ((2, 8), (1, 8))):
                add(dtype, 0, vec, 2, 0, kmax, depth, pair=1, prod=1 if depth == (4 if dtype == 1 else 8) else 0)
        # contiguous equal-width bins: per-lane LDS histogram (direct-load path only)
        for (vec, depth) in (((1, 4),) if dtype == 1 else ((1, 8), (2, 8))):
            for stat in (0, 1):
                for sl in (0, 1):
                    add(dtype, 0, vec, stat, 16, 16, depth, tki=1, sl=sl, hb=1)
                    if vec == 1:    # exactly representable edges: computed instead of read from the LDS table
                        add(dtype, 0, vec, stat, 16, 16, depth, tki=1, sl=sl, hb=1, ha=1)
        # tuning arms for the headline shapes (scripts/kbench.py)
        for (stat, nthr, kmax) in headline:
            for depth in (4, 8, 16):
                add(dtype, 1, vec16[dtype], stat, nthr, kmax, depth, prod=0)
            add(dtype, 1, vec16[dtype], stat, nthr, kmax, 4, nt=0, prod=0)
            for depth in (4, 8):
                add(dtype, 0, 1, stat, nthr, kmax, depth, prod=0)
                add(dtype, 0, 2, stat, nthr, kmax, depth, prod=0)
                if dtype == 0:
                    add(dtype, 0, 4, stat, nthr, kmax, depth, prod=0)
    # round 3 arms (appended, so that the translation units above keep their contents): the arithmetic-edge histogram with
    # deeper bursts and with two cells per lane — the C4 kernel is short of bytes in flight (profiles/r03_c4_bound_pmc.txt)
    for dtype in (() if kind == "dev" else (0, 1)):
        for (vec, depth) in (((1, 8), (1, 16), (2, 4), (2, 8)) if dtype == 1 else ((1, 12), (1, 16), (1, 24), (2, 8), (2, 16))):
            add(dtype, 0, vec, 0, 16, 16, depth, tki=1, sl=1, hb=1, ha=1, prod=0)
        # f32, two cells per lane, FOUR rows in flight: +2.6 % over eight on multi-column plans on large grids (configs[1] on float32
        # storage: 6.31 vs 6.15 TB/s, profiles/r03_sweep_chunks_depth.txt); small grids keep eight (5.73 vs 5.41)
        if dtype == 0:
            for (stat, nthr, kmax) in [(st, nt_, km) for st in (0, 1, 3) for nt_ in (0, 1) for km in (2, 6) if not (st == 0 and nt_ == 0)] + [(2, 0, 2), (2, 0, 6)]:
                add(dtype, 0, 2, stat, nthr, kmax, 4)
        # (tmin, tmax) pairs with the lean group end (FEAT bit 8): sine_dd / min / max sources (stat 2) and mean / sum alone (stat 1)
        for (vec, depth) in (((1, 4), (1, 8)) if dtype == 1 else ((2, 8), (1, 8), (2, 4))):
            prod = 1 if (vec, depth) in ((1, 4), (2, 8), (1, 8)) and not (dtype == 1 and depth == 8) else 0
            add(dtype, 0, vec, 2, 0, 2, depth, pair=1, ss=2, prod=prod)      # every column a plain sine_dd (configs[4])
            add(dtype, 0, vec, 2, 0, 2, depth, pair=1, ss=1, prod=prod)
            add(dtype, 0, vec, 2, 0, 6, depth, pair=1, ss=1, prod=prod)
            add(dtype, 0, vec, 1, 0, 2, depth, pair=1, ss=1, prod=prod)
            add(dtype, 0, vec, 1, 0, 6, depth, pair=1, ss=1, prod=prod)
        # (four cells per lane — half the per-wave scalar work per cell — measured level with two: 4.28 vs 4.25 ms on C5; not kept)
        # inner groups of exactly four rows (6-hourly data): the lean short-group form (FEAT bit 10)
        for (vec, depth) in (((1, 8), (1, 4)) if dtype == 1 else ((2, 8), (1, 8))):
            prod = 0 if (dtype == 1 and depth == 4) else 1
            for stat in (1, 2):
                for kmax in (2, 6):
                    add(dtype, 0, vec, stat, 0, kmax, depth, pair=1, ss=1, quad=1, prod=prod)
        # inner groups of exactly three rows (8-hourly data): the same lean form (FEAT bit 12), two groups per block of six rows
        for vec in ((1,) if dtype == 1 else (2, 1)):
            for stat in (1, 2):
                for kmax in (2, 6):
                    add(dtype, 0, vec, stat, 0, kmax, 6, pair=1, ss=1, tri=1)
    # region-fused period ends (FEAT bit 11): twins of the production two-level variants on the direct-load path.  Round 3 built the
    # twins that gain from two periods on (up to six columns and four threshold slots, with a statistic; of the short-group forms
    # every lean four-row form and the six-column lean pair form); round 4 adds threshold-only plans (a daily panel of degree days) and
    # every short-group form incl. the sine-only pair form (afhip_api.hip: rf_plan_ok says when the planner takes them).  Plans of
    # more than six columns or four threshold slots are bound by their arithmetic and measured level with or behind the per-cell route
    # (13 degree-day columns, daily panel: 20.7 against 21.1 ms; monthly: 15.5 against 14.5): no twins.  Single-level (`sl`),
    # integer-bin and histogram variants have none either.
    # — appended, so that the translation units above keep their contents
    for v in list(out):
        dtype, pipe, vec, stat, nthr, kmax, depth, feat, prod = v
        if prod and pipe == 0 and kmax <= 6 and nthr <= 4 and not (feat & (8 | 16 | 32)):
            out.append((dtype, pipe, vec, stat, nthr, kmax, depth, feat | 2048, prod))
    # inner groups of MIXED lengths one to four rows (FEAT bit 13; a sub-daily series with missing steps): the four-row form with a
    # scalar trip count per group, and its region-fused twins — appended likewise
    if kind != "dev":
        for dtype in (0, 1):
            for vec in ((1,) if dtype == 1 else (2, 1)):
                for stat in (1, 2):
                    for kmax in (2, 6):
                        add(dtype, 0, vec, stat, 0, kmax, 8, pair=1, ss=1, rag=1)
                        add(dtype, 0, vec, stat, 0, kmax, 8, pair=1, ss=1, rag=1, rf=1)
    # `full` = what the planner can pick; the tuning arms of the headline shapes (kbench.py / r03_arms.py `tuning=`; not production:
    # 74 kernels) are compiled by `make MENU=arms` only
    if kind != "arms":
        out = [v for v in out if v[8]]
    return out


def name_of(v):
    dtype, pipe, vec, stat, nthr, kmax, depth, feat, prod = v
    return (f"{'f32' if dtype == 0 else 'f64'}_p{pipe}_v{vec}_s{stat}_t{nthr}_k{kmax}_d{depth}" + ("_nt" if feat & 4 else "")
            + ("_ibins" if feat & 8 else "") + ("_sl" if feat & 16 else "") + ("_hist" if feat & 32 else "") + ("_arith" if feat & 64 else "") + ("_pair" if feat & 128 else "") + ("_ss" if feat & 512 else ("_lean" if feat & 256 else "")) + ("_quad" if feat & 1024 else "") + ("_tri" if feat & 4096 else "") + ("_rag" if feat & 8192 else "") + ("_rf" if feat & 2048 else ""))


def inst(v):
    dtype, pipe, vec, stat, nthr, kmax, depth, feat, prod = v
    return f"k_fused_temporal<{CT[dtype]}, {pipe}, {vec}, {stat}, {nthr}, {kmax}, {max(depth, 1)}, {feat}>"


class _KeepIfSame:
    """Write a file only when its content changes, so `make` does not recompile an unchanged menu."""

    def __init__(self, fn):
        self.fn, self.parts = fn, []

    def __enter__(self):
        return self

    def write(self, text):
        self.parts.append(text)

    def __exit__(self, *exc):
        new = "".join(self.parts)
        try:
            with open(self.fn) as f:
                if f.read() == new:
                    return False
        except OSError:
            pass
        with open(self.fn, "w") as f:
            f.write(new)
        return False


def main():
    outdir = sys.argv[1]
    kind = "full"
    per_file = 4
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == "--menu":
            kind = args.pop(0)
        elif a == "--per-file":
            per_file = int(args.pop(0))
    os.makedirs(outdir, exist_ok=True)
    vs = menu(kind)
    files = []
    ngroups = 0
    for i in range(0, len(vs), per_file):
        group = vs[i:i + per_file]
        idx = i // per_file
        ngroups += 1
        fn = os.path.join(outdir, f"variants_{idx:02d}.hip")
        with _KeepIfSame(fn) as f:
            f.write("// generated by gen_variants.py — do not edit\n")
            f.write('#include "afhip_kernels.h"\n#include "afhip_variants.h"\n')
            f.write("namespace afhip {\n")
            for v in group:
                f.write(f"template __global__ void {inst(v)}(const FusedArgs);\n")
            # host-only registration: kernel handles are taken in the TU that defines them
            f.write(f"int register_variants_{idx:02d}(Variant* out) {{\n    int n = 0;\n")
            for v in group:
                dtype, pipe, vec, stat, nthr, kmax, depth, feat, prod = v
                f.write(f"    out[n++] = Variant{{{dtype}, {pipe}, {vec}, {stat}, {nthr}, {kmax}, {depth}, {1 if feat & 4 else 0}, {prod}, {1 if feat & 8 else 0}, {1 if feat & 16 else 0}, {1 if feat & 32 else 0}, {1 if feat & 64 else 0}, {1 if feat & 128 else 0}, {2 if feat & 512 else (1 if feat & 256 else 0)}, {1 if feat & 1024 else (2 if feat & 4096 else (3 if feat & 8192 else 0))}, {1 if feat & 2048 else 0}, (const void*)&{inst(v)}, \"{name_of(v)}\"}};\n")
            f.write("    return n;\n}\n}\n")
        files.append(fn)
    with _KeepIfSame(os.path.join(outdir, "variants_table.hip")) as f:
        f.write("// generated by gen_variants.py — do not edit\n")
        f.write('#include "afhip_variants.h"\n')
        f.write("namespace afhip {\n")
        for g in range(ngroups):
            f.write(f"int register_variants_{g:02d}(Variant* out);\n")
        f.write(f"static Variant g_table[{len(vs)}];\nstatic int g_count = -1;\n")
        f.write(f'const char* variants_menu() {{ return "{kind}"; }}\n')
        f.write("const Variant* variants_table(int* n) {\n    if (g_count < 0) {\n        int c = 0;\n")
        for g in range(ngroups):
            f.write(f"        c += register_variants_{g:02d}(g_table + c);\n")
        f.write("        g_count = c;\n    }\n    *n = g_count;\n    return g_table;\n}\n}\n")
    print(" ".join(os.path.basename(x) for x in files + [os.path.join(outdir, 'variants_table.hip')]))


if __name__ == "__main__":
    main()
