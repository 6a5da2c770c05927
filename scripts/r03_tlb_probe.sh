#!/bin/bash
# Does a slow plan INSTANCE (scripts/probe/plan_order.py) miss more in the address-translation caches?  Kernel trace + UTCL1 counters over
# N identical plans of a multi-period shape; the post-processing pairs every k_fused_temporal dispatch with its plan (launch order).
o=gpurun_out/r03; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--plan ref --dtype f32 --T 8760 --ny 721 --nx 1440 --spd 24 --periods 12 --regions 3100 --n 6 --rounds 3 --short"
python3 scripts/probe/plan_order.py $args > $o/tlb_probe_plain.log 2>&1; grep -E "forward|reversed" $o/tlb_probe_plain.log
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_LFIFO_FULL TCP_UTCL1_STALL_INFLIGHT_MAX"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $o/tlb_$n -o p -- python3 scripts/probe/plan_order.py $args > $o/tlb_$n.log 2>&1 || echo "pass $n failed"
  grep -E "forward|reversed" $o/tlb_$n.log
done
python3 - <<PY
import pandas as pd, glob
for d in sorted(glob.glob("$o/tlb_TCP*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no counters"); continue
    c = pd.read_csv(f[0])
    c = c[c["Kernel_Name"].str.contains("k_fused_temporal")]
    t = c.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum").sort_index().reset_index(drop=True)
    kt = glob.glob(d + "**/*kernel_trace.csv", recursive=True)
    if kt:
        k = pd.read_csv(kt[0]); k = k[k["Kernel_Name"].str.contains("k_fused_temporal")].sort_values("Dispatch_Id").reset_index(drop=True)
        t["ms"] = (k["End_Timestamp"] - k["Start_Timestamp"]) / 1e6
    t["plan"] = [i % 6 for i in range(len(t))]
    print(d); print(t.head(24).to_string())
PY
