#!/bin/bash
# round 4: the daily panel's chunk count under the scan form of the period end (every workgroup derives its run tables at start: fewer, longer chunks?)
# + 8-hourly (three-row lean form) against 6-hourly (four-row) on the 0.1 deg grid
set -e
mkdir -p gpurun_out/r04
out=gpurun_out/r04/daily_chunks.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{|max rel|^[A-Za-z_0-9=,]+: variant' | cut -c1-330 | tee -a $out; }
run --plan c2 --dtype f32 --periods 365 --arms base AFHIP_NO_PERIOD_CHUNKS=1 AFHIP_WGS_PER_CU=8 AFHIP_WGS_PER_CU=16 AFHIP_WGS_PER_CU=32 AFHIP_WGS_PER_CU=64
run --plan c2 --dtype f64 --periods 365 --arms base AFHIP_NO_PERIOD_CHUNKS=1 AFHIP_WGS_PER_CU=8 AFHIP_WGS_PER_CU=16 AFHIP_WGS_PER_CU=32
run --plan c2 --dtype f32 --periods 73 --arms base AFHIP_NO_PERIOD_CHUNKS=1 AFHIP_WGS_PER_CU=8 AFHIP_WGS_PER_CU=16
run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12 --arms base AFHIP_NO_PERIOD_CHUNKS=1
out=gpurun_out/r04/three_row_groups.txt
: > $out
run --plan meanpoly --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --periods 1 --regions 40000 --arms base AFHIP_NO_QUAD_MODE=1
run --plan meanpoly --dtype f32 --T 1095 --ny 1801 --nx 3600 --spd 3 --periods 1 --regions 40000 --arms base AFHIP_NO_QUAD_MODE=1
run --plan meanpoly --dtype f64 --T 1095 --ny 721 --nx 1440 --spd 3 --periods 1 --arms base AFHIP_NO_QUAD_MODE=1
run --plan meanpoly --dtype f32 --T 1095 --ny 721 --nx 1440 --spd 3 --periods 12 --arms base AFHIP_NO_REGION_FUSED=1 AFHIP_NO_QUAD_MODE=1
run --plan mean --dtype f32 --T 1095 --ny 1801 --nx 3600 --spd 3 --periods 1 --regions 40000 --arms base AFHIP_NO_QUAD_MODE=1
