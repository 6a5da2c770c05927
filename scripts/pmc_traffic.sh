#!/bin/bash
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE; separate --pmc passes) of the engine kernels for one kbench workload.
#   scripts/pmc_traffic.sh <tag> <kbench args...>   -> gpurun_out/pmc_traffic_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pt_${tag}_$c -o p -- python3 scripts/kbench.py "$@" --tunings 0 --rounds 2 > gpurun_out/pt_${tag}_$c.log 2>&1 || echo "pass $c failed"
done
python3 - "$tag" <<'PY'
import sys, glob, pandas as pd
tag = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/pt_{tag}_*/**/*counter_collection.csv", recursive=True):
    d = pd.read_csv(f)
    d = d[d["Kernel_Name"].str.contains("afhip::")]
    d["Kernel_Name"] = d["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void afhip::", "").str.replace("afhip::", "")
    rows.append(d.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean().reset_index())
t = pd.concat(rows).pivot_table(index="Kernel_Name", columns="Counter_Name", values="Counter_Value")
# MI355X_MICROARCH.md (HBM / rocprofv3): counters are in KiB; gfx950 FETCH_SIZE under-reports streaming reads by 2x
t["read_GB"] = t["FETCH_SIZE"] * 2 * 1024 / 1e9
t["write_GB"] = t["WRITE_SIZE"] * 1024 / 1e9
print(t.to_string())
PY
