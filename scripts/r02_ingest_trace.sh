#!/bin/bash
# Where the store -> HBM time goes with the chunk decode on the GPU / on the host: host phases (AGGFLY_HIP_INGEST_TRACE) and kernels (rocprofv3).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02; mkdir -p $o
for m in 1 0; do
  export AGGFLY_HIP_GPU_DECODE=$m AGGFLY_HIP_INGEST_TRACE=1
  python3 scripts/e2e_bench.py > $o/trace_e2e_$m.log 2>&1; grep "ingest trace" $o/trace_e2e_$m.log | tail -2; grep -E "open_decode|decoded_GBps" $o/trace_e2e_$m.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/rp_e2e_$m -o p -- python3 scripts/e2e_bench.py > $o/rp_e2e_$m.log 2>&1
  python3 - $o/rp_e2e_$m <<'PY'
import sys, glob, pandas as pd
d = pd.read_csv(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])
d["Name"] = d["Name"].str.replace(r"\(.*", "", regex=True).str.slice(0, 60)
print(d.sort_values("TotalDurationNs", ascending=False)[["Name", "Calls", "TotalDurationNs", "AverageNs"]].head(8).to_string(index=False))
PY
done
