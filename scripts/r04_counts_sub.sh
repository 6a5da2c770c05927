#!/bin/bash
# round 4: the packed-count gather with a group of lanes per (row, period) pair, pairs dealt period-major (k_csr_spmm_counts_sub), against one lane per pair
mkdir -p gpurun_out/r04
o=gpurun_out/r04/counts_sub_raw.txt; : > $o
run() { echo "== $*" >> $o; timeout -k 10 200 python scripts/r03_arms.py "$@" >> $o 2>&1; }
A="base AFHIP_COUNTS_SPMM_SUB=4 AFHIP_COUNTS_SPMM_SUB=8 AFHIP_COUNTS_SPMM_SUB=16"
run --plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600 --arms $A
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 365 --arms $A
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 12 --arms $A
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 1 --arms $A
run --plan c4 --dtype f32 --T 8760 --ny 1801 --nx 360 --spd 1 --periods 52 --regions 40000 --arms $A
run --plan c4 --dtype f32 --T 8760 --ny 180 --nx 288 --spd 1 --periods 52 --regions 60 --arms $A
python scripts/r04_fmt_arms.py $o > gpurun_out/r04/counts_sub.txt; grep -v "^$" gpurun_out/r04/counts_sub.txt
