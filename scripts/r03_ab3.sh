#!/bin/bash
# three library builds on one box over the short-group shapes
set -e
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03; mkdir -p $o
run() { # tag args...
  tag=$1; shift
  for rep in 1 2; do
    for L in _ab/lib_head.so _ab/lib_semi.so _ab/lib_ring.so; do
      n=$(basename $L .so)
      AGGFLY_HIP_LIB=$PWD/$L python scripts/r03_arms.py "$@" > $o/ab3_${tag}_${n}_$rep.log 2>&1
      echo "$tag $n rep$rep: $(grep -E '^\{' $o/ab3_${tag}_${n}_$rep.log | sed 's/"sequence.*//' | cut -c1-170 | tr '\n' ' ')"
    done
  done
}
run pairpoly_f32 --plan meanpoly --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 7 --arms base
run pairpoly_f64 --plan meanpoly --dtype f64 --ny 1801 --nx 3600 --T 366 --spd 2 --regions 40000 --rounds 7 --arms base
run quadpoly_f32 --plan meanpoly --dtype f32 --ny 721 --nx 1440 --T 1460 --spd 4 --rounds 7 --arms base
run quadmean_f64 --plan mean --dtype f64 --ny 721 --nx 1440 --T 1460 --spd 4 --rounds 7 --arms base
