#!/bin/bash
# round 4: k_rf_reduce's order over the (region, period) pairs (AFHIP_RF_REDUCE_ORDER=r|p) under both layouts of the run sums
mkdir -p gpurun_out/r04
out=gpurun_out/r04/batch7.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" --arms AFHIP_RF_LAYOUT=slot,AFHIP_RF_REDUCE_ORDER=r AFHIP_RF_LAYOUT=slot,AFHIP_RF_REDUCE_ORDER=p AFHIP_RF_LAYOUT=run,AFHIP_RF_REDUCE_ORDER=r base 2>&1 | grep -E '^\{|max rel' | cut -c1-330 | tee -a $out; }
run --plan c2 --dtype f32 --periods 365
run --plan c2 --dtype f32 --periods 73
run --plan c2 --dtype f64 --periods 365
run --plan dd --dtype f32 --periods 365
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 365 --regions 40000
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 52 --regions 40000
run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 365
run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12
