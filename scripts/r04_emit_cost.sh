#!/bin/bash
# round 4: what does a region-fused period end cost in instructions?  VALU / SALU / LDS instruction counts of the configs[1] f32 kernel at 1 and 365
# periods (region-fused twin, per-cell twin) — the difference over 883 k period ends (2,419 wave tiles x 365)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04/emit_cost
for P in 1 73 365; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d ${o}_P$P -o p -- \
    python3 scripts/r03_arms.py --plan c2 --dtype f32 --periods $P --rounds 2 --arms base AFHIP_NO_REGION_FUSED=1 > ${o}_P$P.log 2>&1 || echo "pass P=$P failed"
done
python3 - <<'PY'
import glob, pandas as pd
for P in (1, 73, 365):
    fs = glob.glob(f"gpurun_out/r04/emit_cost_P{P}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counters for P =", P); continue
    d = pd.read_csv(fs[0])
    d = d[d["Kernel_Name"].str.contains("k_fused_temporal")]
    d["k"] = d["Kernel_Name"].str.replace(r"void afhip::", "", regex=False).str.slice(0, 60)
    t = d.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
    print(f"== P = {P}: mean per launch"); print(t.to_string())
PY
