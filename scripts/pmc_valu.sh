#!/bin/bash
# VALU / SALU instruction counts of the fused temporal kernel for a few plans (one --pmc pass each).
#   scripts/pmc_valu.sh   -> gpurun_out/pmc_valu.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/pv_$tag -o p -- python3 scripts/kbench.py "$@" --tunings 0 --rounds 2 > gpurun_out/pv_$tag.log 2>&1
  python3 - "$tag" "$@" <<'PY'
import sys, glob, pandas as pd
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/pv_{tag}/**/*counter_collection.csv", recursive=True)[0]
d = pd.read_csv(f)
d = d[d["Kernel_Name"].str.contains("k_fused_temporal")]
g = d.groupby("Counter_Name")["Counter_Value"].mean()
print(tag, " ".join(sys.argv[2:]), "|", d["Kernel_Name"].iloc[0].split("(")[0], "| VALU", g.get("SQ_INSTS_VALU"), "SALU", g.get("SQ_INSTS_SALU"))
PY
}
run f32_c1 --plan c1 --dtype f32
run f32_c2 --plan c2 --dtype f32
run f64_c1 --plan c1 --dtype f64
run f64_c2 --plan c2 --dtype f64
