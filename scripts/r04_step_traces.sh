#!/bin/bash
# round 4: what a whole bench.py step of each other_configs shape runs on the closing build — rocprofv3 --kernel-trace --stats over
# `AGGFLY_BENCH_ONLY=<shape> python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest`, engine kernels only (name, calls, average ns, total ns)
o=gpurun_out/r04; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$o/step_traces.txt; : > $out
for cfg in C1 C3 C4 C5 C5_iid DAILY REF; do
  export AGGFLY_BENCH_ONLY=$cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/rp_cfg_$cfg -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > $o/rp_cfg_$cfg.log 2>&1
  echo "== $cfg" >> $out
  python3 - >> $out <<PY
import pandas as pd, glob
f = glob.glob("$o/rp_cfg_$cfg/**/*kernel_stats.csv", recursive=True)[0]
d = pd.read_csv(f); d["Name"] = d["Name"].str.replace(r"\(.*", "", regex=True).str.slice(0, 90)
d = d[d["Name"].str.contains("afhip") & ~d["Name"].str.contains("lz4|unshuffle|place_box|read_probe")]
big = d[d["Calls"] >= 10]            # the shape's own launches (10 timed + warm-up); the headline's three run 3 times
print(big[["Name", "Calls", "AverageNs", "TotalDurationNs"]].head(8).to_string(index=False, header=False))
PY
done
cat $out
