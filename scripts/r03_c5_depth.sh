#!/bin/bash
# C5 ring depth A/B (one process, interleaved): scripts/r03_c5_depth.sh TAG
set -e
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03; mkdir -p $o
S="--plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 9"
python scripts/r03_arms.py $S --data era5 --out $o/c5_depth_$1.json --arms base tuning=206 tuning=204 "tuning=206,AFHIP_WGS_PER_CU=8" "tuning=204,AFHIP_WGS_PER_CU=8" base > $o/c5_depth_$1.log 2>&1
grep -E '^\{' $o/c5_depth_$1.log | cut -c1-250
