#!/bin/bash
# round 4: where a group of lanes per (row, period) pair beats one lane in the packed-count gather: entries per row x periods (215 x 1440, hourly year, 13 bins)
mkdir -p gpurun_out/r04
o=gpurun_out/r04/counts_sub2_raw.txt; : > $o
run() { echo "== $*" >> $o; timeout -k 10 200 python scripts/r03_arms.py "$@" >> $o 2>&1; }
A="base AFHIP_COUNTS_SPMM_SUB=4 AFHIP_COUNTS_SPMM_SUB=8 AFHIP_COUNTS_SPMM_SUB=16"
for R in 12000 6000 3100 1000; do for P in 1 12 52 120; do
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods $P --regions $R --rounds 5 --arms $A
done; done
python scripts/r04_fmt_arms.py $o > gpurun_out/r04/counts_sub2.txt; grep -v "^$\|max rel\|\[f32" gpurun_out/r04/counts_sub2.txt | cut -c1-150
