#!/bin/bash
# rocprofv3 --kernel-trace --stats of scripts/kbench.py on every BASELINE shape; keeps the afhip rows.
#   bash scripts/rocprof_shapes.sh   (on the GPU box; writes gpurun_out/rocprof_<tag>.csv)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_$tag -o p -- python3 scripts/kbench.py "$@" --tunings 0 --rounds 5 > gpurun_out/rp_$tag.log 2>&1
  python3 - "$tag" <<'PY'
import sys, pandas as pd
tag = sys.argv[1]
d = pd.read_csv(f"gpurun_out/rp_{tag}/p_kernel_stats.csv")
d = d[d.Name.str.contains("afhip")].copy()
d["Name"] = d["Name"].str.replace(r"\(.*", "", regex=True)
d.to_csv(f"gpurun_out/rocprof_{tag}.csv", index=False)
print(tag); print(d[["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]].to_string(index=False))
PY
}
run c1_f32 --plan c1 --dtype f32
run c3_f32_40yr --plan c1 --dtype f32 --T 350640 --ny 104 --nx 236 --periods 40
run c4_f32 --plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600
run c5_f32 --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 1 --regions 40000
