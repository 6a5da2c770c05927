#!/bin/bash
# round 4, third batch: period-chunk cap on daily panels; light plans on three-row groups; the fuzzers on the round's build
mkdir -p gpurun_out/r04
out=gpurun_out/r04/batch3.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{|max rel|^[A-Za-z_0-9=,]+: variant' | cut -c1-330 | tee -a $out; }
run --plan c2 --dtype f32 --periods 365 --arms base AFHIP_PERIOD_CHUNK_WGS=524288 AFHIP_PERIOD_CHUNK_WGS=1048576 AFHIP_PERIOD_CHUNK_WGS=131072
run --plan c2 --dtype f64 --periods 365 --arms base AFHIP_PERIOD_CHUNK_WGS=524288 AFHIP_PERIOD_CHUNK_WGS=1048576 AFHIP_PERIOD_CHUNK_WGS=2097152
run --plan dd --dtype f32 --periods 365 --arms base AFHIP_PERIOD_CHUNK_WGS=1048576
run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 365 --regions 40000 --arms base AFHIP_PERIOD_CHUNK_WGS=1048576 AFHIP_PERIOD_CHUNK_WGS=4194304
run --plan c1 --dtype f32 --T 1095 --ny 1801 --nx 3600 --spd 3 --periods 1 --regions 40000 --arms base AFHIP_LEAN_STAT1_MIN_K=3
run --plan mean --dtype f32 --T 1095 --ny 1801 --nx 3600 --spd 3 --periods 1 --regions 40000 --arms base AFHIP_LEAN_STAT1_MIN_K=2
run --plan mean --dtype f64 --T 1095 --ny 721 --nx 1440 --spd 3 --periods 1 --arms base AFHIP_LEAN_STAT1_MIN_K=2
run --plan c1 --dtype f64 --T 1095 --ny 721 --nx 1440 --spd 3 --periods 1 --arms base AFHIP_LEAN_STAT1_MIN_K=3
echo "== fuzz_region_fused seeds 162..461" | tee -a $out
FUZZ_LO=162 FUZZ_HI=462 timeout -k 10 900 python scripts/fuzz_region_fused.py 2>&1 | tail -4 | tee -a $out
echo "== fuzz_more seeds 2044..2543" | tee -a $out
FUZZ_LO=2044 FUZZ_HI=2544 timeout -k 10 900 python scripts/fuzz_more.py 2>&1 | tail -4 | tee -a $out
