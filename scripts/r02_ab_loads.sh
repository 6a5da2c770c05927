#!/bin/bash
# A/B of library builds on every BASELINE shape, two interleaved passes:  scripts/r02_ab_loads.sh OUT lib1.so lib2.so ...
out=$1; shift
mkdir -p gpurun_out/r02
{
for pass in 1 2; do
 for shape in "--plan c2 --dtype f64" "--plan c2 --dtype f32" "--plan c1 --dtype f32" "--plan c1 --dtype f32 --T 350640 --ny 104 --nx 236 --periods 40" "--plan c2 --dtype f64 --ny 104 --nx 236" "--plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600" "--plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000" "--plan c5 --dtype f64 --T 8760 --ny 215 --nx 1440 --spd 24 --periods 1"; do
  for lib in "$@"; do
    echo -n "pass $pass | $lib | $shape :: "
    AGGFLY_HIP_LIB=$PWD/$lib python scripts/kbench.py $shape --tunings 0 --rounds 7 2>&1 | tail -1 | sed -e 's/.*temporal_ms_med": //' -e 's/, "temporal_ms_min.*GBps_med": / ms  /' -e 's/, "GBps_best.*//'
  done
 done
done
} > gpurun_out/r02/$out 2>&1
cat gpurun_out/r02/$out
