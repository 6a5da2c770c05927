#!/bin/bash
# PMC passes over one plan through scripts/r03_arms.py: scripts/r03_pmc_plan.sh TAG CELLSTEPS <r03_arms args...>
tag=$1; cs=$2; shift 2
o=gpurun_out/r03; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python3 scripts/r03_arms.py "$@" --rounds 2 --arms base > gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${tag}_$i.log; }
done
python3 scripts/pmc_merge.py ${tag} > $o/pmc_${tag}.txt 2>&1; tail -22 $o/pmc_${tag}.txt
python3 - <<PY
import pandas as pd
t = pd.read_csv("gpurun_out/pmc_${tag}.csv", index_col=0)
r = t[t.index.str.contains("k_fused_temporal")].iloc[0]
cs = float($cs)
print("per cell-step: VALU", r["SQ_INSTS_VALU"] * 64 / cs, " SALU per 64:", r.get("SQ_INSTS_SALU", float("nan")) * 64 / cs, " SMEM per 64:", r.get("SQ_INSTS_SMEM", float("nan")) * 64 / cs,
      " VALU busy:", r["SQ_ACTIVE_INST_VALU"] / r["SQ_BUSY_CYCLES"] / 8.33)
PY
