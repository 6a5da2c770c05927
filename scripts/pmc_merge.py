#!/usr/bin/env python3
"""Merge the per-pass rocprofv3 counter CSVs of scripts/pmc_pass.sh into one table
(mean per dispatch of each afhip kernel)."""
import glob
import sys

import pandas as pd

tag = sys.argv[1]
rows = []
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True)):
    d = pd.read_csv(f)
    d = d[d["Kernel_Name"].str.contains("afhip::")]
    g = d.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean().reset_index()
    rows.append(g)
if not rows:
    sys.exit("no counter files")
out = pd.concat(rows)
out["Kernel_Name"] = out["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void afhip::", "").str.replace("afhip::", "")
tab = out.pivot_table(index="Kernel_Name", columns="Counter_Name", values="Counter_Value")
tab.to_csv(f"gpurun_out/pmc_{tag}.csv")
pd.set_option("display.width", 250, "display.max_columns", 50)
print(tab.T.to_string())
