#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/batch5.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{|max rel|^[A-Za-z_0-9=,]+: variant' | cut -c1-330 | tee -a $out; }
run --plan c2 --dtype f32 --periods 365 --arms base AFHIP_NO_REGION_FUSED=1
run --plan c2 --dtype f64 --periods 365 --arms base AFHIP_NO_REGION_FUSED=1
run --plan dd --dtype f32 --periods 365 --arms base AFHIP_NO_REGION_FUSED=1
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 365 --regions 40000 --arms base AFHIP_NO_REGION_FUSED=1
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 12 --regions 40000 --arms base AFHIP_FORCE_REGION_FUSED=1
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 52 --regions 40000 --arms base AFHIP_NO_REGION_FUSED=1
run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12 --arms base AFHIP_NO_REGION_FUSED=1
run --plan c2 --dtype f32 --periods 12 --arms base AFHIP_NO_REGION_FUSED=1
run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 12 --arms base AFHIP_NO_REGION_FUSED=1
