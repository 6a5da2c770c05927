#!/bin/bash
# Round 3, C5 sine_dd kernel: parity tests, accuracy, kbench (iid / ERA5-like), VALU count per cell-step.  scripts/r03_c5.sh TAG [quick]
tag=${1:-v}
o=gpurun_out/r03; mkdir -p $o
if [ "$2" != "quick" ]; then
  python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -x -q > $o/gputest_$tag.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $o/gputest_$tag.log; tail -3 $o/gputest_$tag.log
  [ $rc -eq 0 ] || exit $rc
fi
python scripts/sine_accuracy.py > $o/sine_accuracy_$tag.json 2>&1; cat $o/sine_accuracy_$tag.json
S="--plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 5"
python scripts/r03_arms.py $S --data era5 --out $o/c5_era5_$tag.json --arms base "tuning=108" "AFHIP_WGS_PER_CU=8" "AFHIP_WGS_PER_CU=16" > $o/c5_era5_$tag.log 2>&1; grep -E '^\{' $o/c5_era5_$tag.log | cut -c1-250
python scripts/r03_arms.py $S --data iid --out $o/c5_iid_$tag.json --arms base > $o/c5_iid_$tag.log 2>&1; grep -E '^\{' $o/c5_iid_$tag.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_r03c5${tag}_$i -o p -- python3 scripts/kbench.py $S --data era5 --tunings 0 --rounds 2 > gpurun_out/pmc_r03c5${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_r03c5${tag}_$i.log; }
done
python3 scripts/pmc_merge.py r03c5${tag} > $o/pmc_c5_$tag.txt 2>&1; tail -25 $o/pmc_c5_$tag.txt
python3 - <<PY
import pandas as pd
t = pd.read_csv("gpurun_out/pmc_r03c5${tag}.csv", index_col=0)
r = t[t.index.str.contains("k_fused_temporal")].iloc[0]
cs = 730 * 1801 * 3600
print("VALU lane-instructions per cell-step:", r["SQ_INSTS_VALU"] * 64 / cs, " LDS:", r.get("SQ_INSTS_LDS", float("nan")) * 64 / cs, " SALU per 64 cell-steps:", r.get("SQ_INSTS_SALU", float("nan")) * 64 / cs)
PY
