#!/bin/bash
# round 4: (1) where the plans' scratch comes from — library hipMalloc against torch's caching allocator (the new default) — on the configs[3] shape;
# (2) the slot gather's order over the grid (AFHIP_SLOT_SPMM_ORDER=v|p) on the per-cell route at 12 / 73 / 365 periods and K = 13
set -e
mkdir -p gpurun_out/r04
out=gpurun_out/r04/plan_order_probe.txt
: > $out
for ws in mix library torch; do
  echo "== --ws $ws" | tee -a $out
  timeout -k 10 300 python scripts/probe/plan_order.py --short --rounds 7 --n 4 --ws $ws 2>&1 | grep -E "scratch|forward|reversed|variant" | cut -c1-300 | tee -a $out
done
out=gpurun_out/r04/slot_order.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{|max rel' | cut -c1-260 | tee -a $out; }
for P in 12 73 365; do
  run --plan c2 --dtype f32 --periods $P --arms AFHIP_NO_REGION_FUSED=1,AFHIP_SLOT_SPMM_ORDER=v AFHIP_NO_REGION_FUSED=1,AFHIP_SLOT_SPMM_ORDER=p base
done
run --plan c2 --dtype f64 --periods 365 --arms AFHIP_NO_REGION_FUSED=1,AFHIP_SLOT_SPMM_ORDER=v AFHIP_NO_REGION_FUSED=1,AFHIP_SLOT_SPMM_ORDER=p base
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 12 --regions 40000 --arms AFHIP_SLOT_SPMM_ORDER=v AFHIP_SLOT_SPMM_ORDER=p
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 365 --regions 40000 --arms AFHIP_SLOT_SPMM_ORDER=v AFHIP_SLOT_SPMM_ORDER=p
run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12 --arms AFHIP_NO_REGION_FUSED=1,AFHIP_SLOT_SPMM_ORDER=v AFHIP_NO_REGION_FUSED=1,AFHIP_SLOT_SPMM_ORDER=p base
