#!/bin/bash
# unnormalised double-double power chain vs the renormalising one: two builds on one box
set -e
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_api.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x -m gpu -k "pow or poly or transform or pair or four_row or c1 or c2 or full" > $o/dd_tests.log 2>&1 || { tail -30 $o/dd_tests.log; exit 1; }
tail -2 $o/dd_tests.log
run() { tag=$1; shift
  for rep in 1 2; do for L in _ab/lib_olddd.so _ab/lib_newdd.so; do n=$(basename $L .so)
      AGGFLY_HIP_LIB=$PWD/$L python scripts/r03_arms.py "$@" > $o/abdd_${tag}_${n}_$rep.log 2>&1
      echo "$tag $n rep$rep: $(grep -E '^\{' $o/abdd_${tag}_${n}_$rep.log | sed 's/"sequence.*//' | cut -c1-170 | tr '\n' ' ')"
  done; done; }
run pairpoly_f32 --plan meanpoly --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 7 --arms base
run pairpoly_f64 --plan meanpoly --dtype f64 --ny 1801 --nx 3600 --T 366 --spd 2 --regions 40000 --rounds 7 --arms base
run quadpoly_f32 --plan meanpoly --dtype f32 --ny 721 --nx 1440 --T 1460 --spd 4 --rounds 7 --arms base
run ref_f32 --plan ref --dtype f32 --ny 721 --nx 1440 --T 8760 --spd 24 --periods 12 --rounds 7 --arms base
run c2_f64 --plan c2 --dtype f64 --ny 215 --nx 1440 --T 8760 --spd 24 --rounds 9 --arms base
run c2_f32 --plan c2 --dtype f32 --ny 215 --nx 1440 --T 8760 --spd 24 --rounds 9 --arms base
