#!/usr/bin/env python3
"""Turns the rocprofv3 passes of scripts/r03_bench_profiles.sh (round 2: r02_bench_profiles.sh; ROUND=r02) into the files bench.py reads:
gpurun_out/<round>/traffic_<tag>.json (entry for profiles/traffic.json: HBM bytes per launch of the headline kernel, gfx950
FETCH_SIZE correction applied, with the library build it was measured on), gpurun_out/r02/valu_counts_<tag>.json
(VALU instructions per grid-cell-timestep of the C5 sine_dd kernel + its issue utilisation) and the kernel stats CSV."""
import glob
import hashlib
import json
import os
import sys

import pandas as pd

tag = sys.argv[1]
rnd = os.environ.get("ROUND", "r03")
o = f"gpurun_out/{rnd}"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id():
    h = hashlib.sha256()
    with open(os.path.join(root, "aggfly_amd", "libaggfly_hip.so"), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:12]


def counters(pattern, kernel_substr):
    rows = {}
    for f in glob.glob(pattern, recursive=True):
        d = pd.read_csv(f)
        d = d[d["Kernel_Name"].str.contains(kernel_substr, regex=False)]
        for name, g in d.groupby("Counter_Name"):
            rows[name] = (float(g["Counter_Value"].mean()), int(len(g)), d["Kernel_Name"].iloc[0].split("(")[0])
    return rows


# ---- kernel stats of the bench command
st = glob.glob(f"{o}/rp_bench_{tag}/**/*kernel_stats.csv", recursive=True)
if st:
    d = pd.read_csv(st[0])
    d = d[d.Name.str.contains("afhip")].copy()
    d["Name"] = d["Name"].str.replace(r"\(.*", "", regex=True)
    d.to_csv(f"{o}/bench_{tag}_kernel_stats.csv", index=False)
    print(d[["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]].to_string(index=False))

# ---- traffic of the headline kernel (configs[1], f64)
T, C = 8760, 215 * 1440
c = {}
for nm in ("FETCH_SIZE", "WRITE_SIZE"):
    c.update(counters(f"{o}/pmc_bench_{tag}_{nm}/**/*counter_collection.csv", "k_fused_temporal<double, 0, 1, 1, 1, 6, 4, 4>"))
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md (HBM / rocprofv3): counters are in KiB; gfx950 FETCH_SIZE reports half the bytes of a coalesced streaming read
    rd, wr = c["FETCH_SIZE"][0] * 2 * 1024, c["WRITE_SIZE"][0] * 1024
    ent = {"kernel": c["FETCH_SIZE"][2].replace("void afhip::", ""), "FETCH_SIZE_KiB": c["FETCH_SIZE"][0], "WRITE_SIZE_KiB": c["WRITE_SIZE"][0],
           "launches_averaged": c["FETCH_SIZE"][1], "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch": T * C * 8, "build": build_id(),
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py itself "
                     "(--steps 5 --no-cpu-baseline --no-other-configs); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the "
                     "bytes of a coalesced streaming read)", "source": f"profiles/{rnd}_pmc_traffic_bench_{tag}.json (scripts/{rnd}_bench_profiles.sh)"}
    json.dump({f"c2_f64_T{T}_C{C}": ent}, open(f"{o}/traffic_{tag}.json", "w"), indent=1)
    print("traffic:", json.dumps(ent, indent=1))

# ---- traffic of the configs[3] kernel (13 bins per year, LDS histogram with arithmetic edges, f32): passes with AGGFLY_BENCH_ONLY=C4
c4 = {}
for nm in ("FETCH_SIZE", "WRITE_SIZE"):
    c4.update(counters(f"{o}/pmc_bench_{tag}_c4_{nm}/**/*counter_collection.csv", "k_fused_temporal<float, 0, 1, 0, 16, 16, 8, "))
if "FETCH_SIZE" in c4 and "WRITE_SIZE" in c4:
    T4, C4 = 91615, 180 * 288
    rd, wr = c4["FETCH_SIZE"][0] * 2 * 1024, c4["WRITE_SIZE"][0] * 1024
    ent = {"kernel": c4["FETCH_SIZE"][2].replace("void afhip::", ""), "FETCH_SIZE_KiB": c4["FETCH_SIZE"][0], "WRITE_SIZE_KiB": c4["WRITE_SIZE"][0],
           "launches_averaged": c4["FETCH_SIZE"][1], "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch": T4 * C4 * 4, "ratio_to_algorithmic": (rd + wr) / (T4 * C4 * 4), "build": build_id(),
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py's own C4 run (AGGFLY_BENCH_ONLY=C4); "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md", "source": f"profiles/{rnd}_pmc_traffic_bench_{tag}.json (scripts/{rnd}_bench_profiles.sh)"}
    try:
        cur = json.load(open(f"{o}/traffic_{tag}.json"))
    except (OSError, ValueError):
        cur = {}
    cur[f"c4_f32_T{T4}_C{C4}"] = ent
    json.dump(cur, open(f"{o}/traffic_{tag}.json", "w"), indent=1)
    print("traffic C4:", json.dumps(ent, indent=1))

# ---- VALU counts of the C5 kernel (the SURVEY 8d field) and of the same kernel on the iid cube (C5_iid: its hostile case)
out_v = {}
for cfg, sub in (("C5", "c5"), ("C5_iid", "c5iid")):
    v = {}
    for f in glob.glob(f"{o}/pmc_bench_{tag}_{sub}_*"):
        if os.path.isdir(f):
            v.update(counters(f"{f}/**/*counter_collection.csv", "2, 0, 2, 8, "))       # the pair-mode sine_dd variant (float, 0, 2, 2, 0, 2, 8, FEAT)
    if "SQ_INSTS_VALU" not in v:
        continue
    T5, C5 = 730, 1801 * 3600
    per = v["SQ_INSTS_VALU"][0] * 64 / (T5 * C5)
    ent = {"valu_inst_per_cell_step": per, "kernel": v["SQ_INSTS_VALU"][2].replace("void afhip::", ""), "build": build_id(),
           "counters_per_launch": {k: x[0] for k, x in v.items()},
           "source": f"profiles/{rnd}_pmc_valu_bench_{tag}.json: SQ_INSTS_VALU x 64 lanes / (T x cells), rocprofv3 --pmc over bench.py's own {cfg} run "
                     f"(AGGFLY_BENCH_ONLY={cfg}, " + ("ERA5-like (tmin, tmax) field)" if cfg == "C5" else "iid cube)")}
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        ent["valu_issue_utilisation"] = v["SQ_ACTIVE_INST_VALU"][0] * 4 / (v["GRBM_GUI_ACTIVE"][0] / 8 * 256 * 4)
    if "SQ_WAIT_INST_ANY" in v and "SQ_WAVE_CYCLES" in v:
        ent["share_of_wave_cycles_waiting"] = v["SQ_WAIT_INST_ANY"][0] / v["SQ_WAVE_CYCLES"][0]
    out_v[cfg] = ent
if out_v:
    json.dump(out_v, open(f"{o}/valu_counts_{tag}.json", "w"), indent=1)
    print("valu:", json.dumps(out_v, indent=1))
