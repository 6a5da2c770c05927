#!/bin/bash
# lean group end with packed column words + written-out power chain vs the previous build (same box)
set -e
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_api.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -x -m gpu > $o/lean2_tests.log 2>&1 || { tail -30 $o/lean2_tests.log; exit 1; }
tail -2 $o/lean2_tests.log
run() { tag=$1; shift
  for rep in 1 2; do for L in _ab/lib_newdd.so _ab/lib_lean2.so; do n=$(basename $L .so)
      AGGFLY_HIP_LIB=$PWD/$L python scripts/r03_arms.py "$@" > $o/ablean_${tag}_${n}_$rep.log 2>&1
      echo "$tag $n rep$rep: $(grep -E '^\{' $o/ablean_${tag}_${n}_$rep.log | sed 's/"sequence.*//' | cut -c1-170 | tr '\n' ' ')"
  done; done; }
run pairpoly_f32 --plan meanpoly --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 7 --arms base
run pairpoly_f64 --plan meanpoly --dtype f64 --ny 1801 --nx 3600 --T 366 --spd 2 --regions 40000 --rounds 7 --arms base
run quadpoly_f32 --plan meanpoly --dtype f32 --ny 721 --nx 1440 --T 1460 --spd 4 --rounds 7 --arms base
run quadpoly_f64 --plan meanpoly --dtype f64 --ny 721 --nx 1440 --T 1460 --spd 4 --rounds 7 --arms base
run c5 --plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 7 --data era5 --arms base
