#!/bin/bash
# round 4: is the placement-dependent speed of the configs[3] kernel (scripts/probe/cube_placement.py: 2.86 ... 3.08 ms from one allocation of the cube to the
# next) visible in the address-translation or L2 counters?  Kernel trace + counters per dispatch, 8 allocations x 14 launches each.
o=gpurun_out/r04; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $o/place_$n -o p -- python3 scripts/probe/cube_placement.py --trials 8 > $o/place_$n.log 2>&1 || echo "pass $n failed"
  grep -E "^trial" $o/place_$n.log | cut -c1-140
done
python3 - <<PY
import pandas as pd, glob
for d in sorted(glob.glob("$o/place_T*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no counters"); continue
    c = pd.read_csv(f[0])
    c = c[c["Kernel_Name"].str.contains("k_fused_temporal")]
    t = c.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum").sort_index().reset_index(drop=True)
    kt = glob.glob(d + "**/*kernel_trace.csv", recursive=True)
    if kt:
        k = pd.read_csv(kt[0]); k = k[k["Kernel_Name"].str.contains("k_fused_temporal")].sort_values("Dispatch_Id").reset_index(drop=True)
        t["ms"] = (k["End_Timestamp"] - k["Start_Timestamp"]) / 1e6
    t["trial"] = [i // 14 for i in range(len(t))]
    g = t[t.trial < 8].groupby("trial").median()
    print(d); print(g.to_string()); print(g.corr()["ms"].to_string())
PY
