#!/bin/bash
# round 4: layout of the region-fused run sums — slot-major [slots][runs][K+1] (round 3) against run-major [runs][slots][K+1]
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rf_layout.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" --arms AFHIP_RF_LAYOUT=slot AFHIP_RF_LAYOUT=run base 2>&1 | grep -E '^\{|max rel|^[A-Za-z_0-9=,]+: variant' | cut -c1-330 | tee -a $out; }
for P in 12 24 73 365; do run --plan c2 --dtype f32 --periods $P; done
for P in 12 365; do run --plan c2 --dtype f64 --periods $P; done
run --plan dd --dtype f32 --periods 365
for P in 52 365; do run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods $P --regions 40000; done
run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12
run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 12
run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 365
