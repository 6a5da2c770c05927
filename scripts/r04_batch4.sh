#!/bin/bash
# round 4, fourth batch: a daily BINS panel from hourly data (packed counts: the per-cell route's remaining many-period form), the API call of a daily panel
mkdir -p gpurun_out/r04
out=gpurun_out/r04/batch4.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{|max rel|^[A-Za-z_0-9=,]+: variant' | cut -c1-330 | tee -a $out; }
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 365 --arms base AFHIP_NO_COUNTS_SPMM=1
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 12 --arms base
echo "== API daily panel" | tee -a $out
timeout -k 10 300 python scripts/r04_api_daily.py 2>&1 | grep -v amdgpu.ids | tee -a $out
echo "== fuzz_region_fused seeds 462..661 (run-major / slot-major layouts by rule)" | tee -a $out
FUZZ_LO=462 FUZZ_HI=662 timeout -k 10 600 python scripts/fuzz_region_fused.py 2>&1 | tail -2 | tee -a $out
