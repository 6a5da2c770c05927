#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03; mkdir -p $o
export AGGFLY_HIP_LIB=$PWD/_ab/lib_semi.so
python scripts/r03_arms.py --plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 9 --data era5 --arms base tuning=206 tuning=204 base > $o/semi_depth_c5.log 2>&1
grep -E '^\{' $o/semi_depth_c5.log | cut -c1-200
python scripts/r03_arms.py --plan meanpoly --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 7 --arms base tuning=206 tuning=204 > $o/semi_depth_poly.log 2>&1
grep -E '^\{' $o/semi_depth_poly.log | cut -c1-200
