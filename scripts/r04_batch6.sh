#!/bin/bash
# A/B of two library builds on one box, alternating processes: the tree against the previous commit (twins at 103 VGPRs: lane weights and words in registers, validity as doubles)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/batch6.txt
: > $out
run() { echo "== [$LIBTAG] $*" | tee -a $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{' | cut -c1-330 | tee -a $out; }
for rep in 1 2; do
for lib in main prev; do
  export LIBTAG=$lib
  if [ $lib = prev ]; then export AGGFLY_HIP_LIB=$PWD/scripts/probe/_build/libaggfly_hip_prev.so; else unset AGGFLY_HIP_LIB; fi
  run --plan c2 --dtype f32 --periods 365 --arms base
  run --plan c2 --dtype f64 --periods 365 --arms base
  run --plan dd --dtype f32 --periods 365 --arms base
  run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 365 --regions 40000 --arms base
  run --plan c2 --dtype f32 --periods 12 --arms base
  run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12 --arms base
done; done
