#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of a store -> HBM read: the last read with the decode on the
GPU is cut out (from its first host -> device copy to its last kernel) and printed as a list of busy intervals per engine."""
import glob, sys
import pandas as pd
d = sys.argv[1]
k = pd.read_csv(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])
m = pd.read_csv(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])
k["name"] = k["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace(r"void afhip::", "", regex=False).str.slice(0, 40)
lz = k[k["name"].str.contains("k_lz4_streams")]
# decode launches come in groups (one read = several batches); a gap of more than 50 ms separates the reads
starts = lz["Start_Timestamp"].values
groups, cur = [], [0]
for i in range(1, len(starts)):
    if starts[i] - starts[i - 1] > 50e6:
        groups.append(cur); cur = []
    cur.append(i)
groups.append(cur)
for gi in (0, len(groups) - 1) if len(groups) > 1 else (0,):    # the first (cold) read, then the last
    g = groups[gi]
    t_lo = starts[g[0]] - 30e6; t_hi = lz["End_Timestamp"].values[g[-1]] + 5e6
    mm = m[(m["Start_Timestamp"] >= t_lo) & (m["End_Timestamp"] <= t_hi) & (m["Direction"].str.contains("HOST_TO_DEVICE"))]
    kk = k[(k["Start_Timestamp"] >= t_lo) & (k["End_Timestamp"] <= t_hi)]
    t0 = min(mm["Start_Timestamp"].min(), kk["Start_Timestamp"].min())
    print(f"--- read {gi}: {len(g)} decode launches; window {(max(kk['End_Timestamp'].max(), mm['End_Timestamp'].max()) - t0) / 1e6:.2f} ms")
    ev = [(r.Start_Timestamp, r.End_Timestamp, "H2D", "") for r in mm.itertuples() if r.End_Timestamp - r.Start_Timestamp > 50e3]
    ev += [(r.Start_Timestamp, r.End_Timestamp, "kernel", r.name) for r in kk.itertuples() if r.End_Timestamp - r.Start_Timestamp > 50e3]
    for s, e, what, name in sorted(ev):
        print(f"  {(s - t0) / 1e6:8.2f} .. {(e - t0) / 1e6:8.2f} ms  {(e - s) / 1e6:7.2f} ms  {what:7s} {name}")
    print(f"  H2D busy {((mm['End_Timestamp'] - mm['Start_Timestamp']).sum()) / 1e6:.2f} ms; kernels busy (sum) {((kk['End_Timestamp'] - kk['Start_Timestamp']).sum()) / 1e6:.2f} ms")
