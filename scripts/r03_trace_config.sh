#!/bin/bash
# kernel trace of one bench.py other_config: scripts/r03_trace_config.sh REF
cfg=$1; o=gpurun_out/r03; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AGGFLY_BENCH_ONLY=$cfg
rocprofv3 --kernel-trace --stats --output-format csv -d $o/rp_cfg_$cfg -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $o/rp_cfg_$cfg.log 2>&1
python3 - <<PY
import pandas as pd, glob
f = glob.glob("$o/rp_cfg_$cfg/**/*kernel_stats.csv", recursive=True)[0]
d = pd.read_csv(f); d["Name"] = d["Name"].str.replace(r"\(.*", "", regex=True).str.slice(0, 90)
print(d[["Name", "Calls", "AverageNs", "TotalDurationNs"]].head(14).to_string(index=False))
PY
