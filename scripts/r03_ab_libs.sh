#!/bin/bash
# A/B of two library builds on ONE box: separate processes, alternating.  scripts/r03_ab_libs.sh TAG libA.so libB.so [r03_arms args...]
set -e
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03; mkdir -p $o
tag=$1; A=$2; B=$3; shift 3
for rep in 1 2 3; do
  for L in $A $B; do
    n=$(basename $L .so)
    AGGFLY_HIP_LIB=$PWD/$L python scripts/r03_arms.py "$@" > $o/ab_${tag}_${n}_$rep.log 2>&1
    echo "$n rep$rep: $(grep -E '^\{' $o/ab_${tag}_${n}_$rep.log | cut -c1-200 | tr '\n' ' ')"
  done
done
