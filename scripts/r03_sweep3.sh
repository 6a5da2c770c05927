#!/bin/bash
# Round 3: chunks per CU x burst depth x workgroup size on every streaming shape (the C4 sweep showed that the number of time
# chunks matters more than round 1's sweep on configs[1] suggested).  scripts/r03_sweep3.sh TAG
tag=${1:-c}
o=gpurun_out/r03; mkdir -p $o
run() { name=$1; shift; python scripts/r03_arms.py "$@" --rounds 7 --out $o/sw3_${name}_$tag.json > $o/sw3_${name}_$tag.log 2>&1; echo "== $name"; grep -E '^\{|unavailable' $o/sw3_${name}_$tag.log | cut -c1-200; }
PC="AFHIP_WGS_PER_CU"
run c2f32 --plan c2 --dtype f32 --arms base "$PC=8" "$PC=16" "$PC=32" "tuning=204" "tuning=204,$PC=8" "tuning=204,$PC=16" "tuning=204,$PC=32" "AFHIP_FORCE_WG=64,$PC=8" "tuning=204,AFHIP_FORCE_WG=64,$PC=8"
run c2f64 --plan c2 --dtype f64 --arms base "$PC=8" "$PC=16" "$PC=32" "AFHIP_FORCE_WG=64" "AFHIP_FORCE_WG=64,$PC=8" "tuning=108" "tuning=108,$PC=8"
run c1f32 --plan c1 --dtype f32 --arms base "$PC=8" "$PC=16" "$PC=32" "tuning=104" "tuning=104,$PC=8" "tuning=204" "AFHIP_FORCE_WG=64,$PC=8"
run c2f32small --plan c2 --dtype f32 --ny 104 --nx 236 --arms base "$PC=8" "$PC=16" "tuning=204" "tuning=204,$PC=8" "tuning=204,$PC=16"
run c2f64small --plan c2 --dtype f64 --ny 104 --nx 236 --arms base "$PC=8" "$PC=16" "$PC=32"
run c3f32 --plan c1 --dtype f32 --T 350640 --ny 104 --nx 236 --periods 40 --arms base "$PC=8" "$PC=16" "$PC=32" "tuning=104" "tuning=104,$PC=16"
run mean6h --plan mean --dtype f32 --ny 721 --nx 1440 --T 1460 --spd 4 --arms base "tuning=204" "tuning=208" "tuning=104" "tuning=108" "tuning=1404" "tuning=1408" "$PC=8" "tuning=204,$PC=8"
run meanpair --plan mean --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --arms base "tuning=204" "tuning=208" "tuning=104" "tuning=108" "tuning=1404" "tuning=1408" "$PC=8"
S4="--plan c4 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600"
run c4f32 $S4 --dtype f32 --arms base "AFHIP_FORCE_WG=64,$PC=32" "AFHIP_FORCE_WG=64,$PC=48" "AFHIP_FORCE_WG=64,$PC=64" "AFHIP_FORCE_WG=64,$PC=96" "tuning=116,$PC=24" "tuning=116,$PC=32" "tuning=116,$PC=48" "tuning=112,$PC=32" "tuning=112,AFHIP_FORCE_WG=64,$PC=32"
run c4f64 $S4 --dtype f64 --arms base "$PC=32" "$PC=48" "$PC=64" "AFHIP_FORCE_WG=64,$PC=8" "AFHIP_FORCE_WG=64,$PC=16" "AFHIP_FORCE_WG=64,$PC=32" "AFHIP_FORCE_WG=64,$PC=64"
