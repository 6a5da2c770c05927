#!/bin/bash
# round 4: the region-fused period end in its scan form (AFHIP_FORCE_REGION_FUSED=1: whatever the planner's rule says) against round 3's
# LDS-staged form (AFHIP_RF_STAGED=1) and the per-cell route (AFHIP_NO_REGION_FUSED=1), on many-period panels and on round 3's shapes
set -e
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rf_wide.txt
: > $out
ARMS="base AFHIP_FORCE_REGION_FUSED=1 AFHIP_FORCE_REGION_FUSED=1,AFHIP_RF_STAGED=1 AFHIP_NO_REGION_FUSED=1"
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" --arms $ARMS 2>&1 | grep -E '^\{|max rel|^base:' | cut -c1-330 | tee -a $out; }
# sine_dd from (tmin, tmax) pairs, 0.1 deg global: monthly / weekly / daily panels
for P in 12 52 365; do run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods $P --regions 40000; done
# degree days alone (stat 0, one slot), hourly counties extent
for P in 12 52 365; do run --plan dd --dtype f32 --periods $P; done
run --plan dd --dtype f64 --periods 365
# 13 degree-day columns (K = 13, 13 slots)
for P in 12 365; do run --plan dd13 --dtype f32 --periods $P; done
# round 3's twins: the configs[1] columns f32 / f64, the reference's benchmark shape, 6-hourly polynomial, pairs polynomial
for P in 12 73 365; do run --plan c2 --dtype f32 --periods $P; done
for P in 12 365; do run --plan c2 --dtype f64 --periods $P; done
run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12
run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 12
run --plan meanpoly --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 12 --regions 40000
