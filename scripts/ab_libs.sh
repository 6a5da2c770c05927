#!/bin/bash
# A/B of library builds in one gpurun call: scripts/ab_libs.sh "<kbench args>" lib1.so lib2.so ...   (two interleaved passes)
args=$1; shift
for pass in 1 2; do
  for lib in "$@"; do
    echo "== pass $pass $lib"
    AGGFLY_HIP_LIB=$PWD/$lib python scripts/kbench.py $args --rounds 7 | grep temporal_ms_med
  done
done
