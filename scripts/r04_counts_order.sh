#!/bin/bash
# round 4: k_csr_spmm_counts dealt region-major (a wave walks the periods of one region: records a slot apart) or period-major (neighbouring regions of one period)
mkdir -p gpurun_out/r04
o=gpurun_out/r04/counts_order_raw.txt; : > $o
run() { echo "== $*" >> $o; timeout -k 10 200 python scripts/r03_arms.py "$@" >> $o 2>&1; }
run --plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600 --arms base AFHIP_COUNTS_SPMM_ORDER=p AFHIP_COUNTS_SPMM_ORDER=r
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 365 --arms base AFHIP_COUNTS_SPMM_ORDER=p
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 12 --arms base AFHIP_COUNTS_SPMM_ORDER=p
run --plan c4 --dtype f32 --T 8760 --spd 1 --periods 2 --arms base AFHIP_COUNTS_SPMM_ORDER=p
run --plan c4 --dtype f64 --T 8760 --spd 1 --periods 365 --arms base AFHIP_COUNTS_SPMM_ORDER=p
run --plan c4 --dtype f32 --T 8760 --ny 1801 --nx 360 --spd 1 --periods 52 --regions 40000 --arms base AFHIP_COUNTS_SPMM_ORDER=p
python scripts/r04_fmt_arms.py $o > gpurun_out/r04/counts_order.txt; grep -v "^$" gpurun_out/r04/counts_order.txt
