#!/bin/bash
# round 4: many-period panels with ONE cell per lane (tuning=108: fewer registers, twice the period ends per cell, each without the two-cell joins) against the planner's two
mkdir -p gpurun_out/r04
o=gpurun_out/r04/daily_vec_raw.txt; : > $o
run() { echo "== $*" >> $o; timeout -k 10 200 python scripts/r03_arms.py "$@" >> $o 2>&1; }
run --plan c2 --dtype f32 --periods 365 --arms base tuning=108 tuning=208 tuning=104
run --plan c2 --dtype f32 --periods 73 --arms base tuning=108
run --plan c2 --dtype f32 --periods 12 --arms base tuning=108
run --plan dd --dtype f32 --periods 365 --arms base tuning=108
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 365 --regions 40000 --arms base tuning=108
run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 365 --arms base tuning=108
python scripts/r04_fmt_arms.py $o > gpurun_out/r04/daily_vec.txt; grep -v "^$" gpurun_out/r04/daily_vec.txt; grep "^tuning" $o | cut -c1-160
