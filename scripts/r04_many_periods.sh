#!/bin/bash
# round 4: where do many-period panels stand? (per-cell route vs region-fused), configs[1] columns at P = 365 / 73, K = 8 columns, monthly sine pairs
set -e
out=gpurun_out/r04/many_periods.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{|^base|variant=' | cut -c1-400 | tee -a $out; }
run --plan c2 --dtype f32 --periods 365 --arms base AFHIP_NO_REGION_FUSED=1
run --plan c2 --dtype f64 --periods 365 --arms base AFHIP_NO_REGION_FUSED=1
run --plan c2 --dtype f32 --periods 73 --arms base AFHIP_NO_REGION_FUSED=1
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 12 --regions 40000 --arms base AFHIP_FORCE_REGION_FUSED=1
run --plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600 --arms base AFHIP_X=1
