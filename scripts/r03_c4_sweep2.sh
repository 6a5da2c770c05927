#!/bin/bash
# Round 3, C4: bytes in flight — burst depth x chunks per CU x cells per lane, both storage types.  scripts/r03_c4_sweep2.sh TAG
tag=${1:-b}
o=gpurun_out/r03; mkdir -p $o
S="--plan c4 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600 --data era5 --rounds 7"
python scripts/r03_arms.py $S --dtype f32 --out $o/c4_arms2_f32_$tag.json --arms base "AFHIP_WGS_PER_CU=24" "AFHIP_WGS_PER_CU=32" "AFHIP_WGS_PER_CU=48" "AFHIP_WGS_PER_CU=64" \
   "AFHIP_FORCE_WG=64,AFHIP_WGS_PER_CU=16" "AFHIP_FORCE_WG=64,AFHIP_WGS_PER_CU=24" "AFHIP_FORCE_WG=64,AFHIP_WGS_PER_CU=32" \
   "tuning=112" "tuning=116" "tuning=124" "tuning=116,AFHIP_WGS_PER_CU=8" "tuning=116,AFHIP_WGS_PER_CU=16" "tuning=116,AFHIP_WGS_PER_CU=24" "tuning=124,AFHIP_WGS_PER_CU=16" \
   "tuning=208" "tuning=216" "tuning=208,AFHIP_WGS_PER_CU=16" "tuning=216,AFHIP_WGS_PER_CU=16" "tuning=216,AFHIP_WGS_PER_CU=24" "tuning=116,AFHIP_FORCE_WG=64,AFHIP_WGS_PER_CU=16" \
   > $o/c4_arms2_f32_$tag.log 2>&1; grep -E '^\{|unavailable|diff' $o/c4_arms2_f32_$tag.log
python scripts/r03_arms.py $S --dtype f64 --out $o/c4_arms2_f64_$tag.json --arms base "AFHIP_WGS_PER_CU=16" "AFHIP_WGS_PER_CU=24" "AFHIP_WGS_PER_CU=32" \
   "tuning=108" "tuning=116" "tuning=108,AFHIP_WGS_PER_CU=16" "tuning=108,AFHIP_WGS_PER_CU=24" "tuning=116,AFHIP_WGS_PER_CU=16" "tuning=204" "tuning=208" "tuning=204,AFHIP_WGS_PER_CU=16" "tuning=208,AFHIP_WGS_PER_CU=16" \
   > $o/c4_arms2_f64_$tag.log 2>&1; grep -E '^\{|unavailable|diff' $o/c4_arms2_f64_$tag.log
