#!/bin/bash
# A/B, alternating processes: the general float32 two-cell twins with a slim period end (one column per block, weights re-read per block, store addresses formed at the stores:
# 79 VGPRs = six waves per SIMD) against 89 VGPRs / five waves (scripts/probe/_build/libaggfly_hip_prev.so)
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "region_fused or launch_shape or many_period" > gpurun_out/r04/batch12_tests.log 2>&1; tail -1 gpurun_out/r04/batch12_tests.log
out=gpurun_out/r04/batch12.txt
: > $out
run() { echo "== [$LIBTAG] $*" >> $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{' | cut -c1-330 >> $out; }
for rep in 1 2 3; do
for lib in main prev; do
  export LIBTAG=$lib
  if [ $lib = prev ]; then export AGGFLY_HIP_LIB=$PWD/scripts/probe/_build/libaggfly_hip_prev.so; else unset AGGFLY_HIP_LIB; fi
  run --plan c2 --dtype f32 --periods 365 --arms base
  run --plan c2 --dtype f32 --periods 73 --arms base
  run --plan c2 --dtype f32 --periods 12 --arms base
  run --plan ref --dtype f32 --ny 721 --nx 1440 --periods 12 --arms base
  run --plan meanpoly --dtype f32 --T 8760 --periods 365 --arms base
  run --plan c1 --dtype f32 --T 8760 --periods 52 --arms base
done; done
python3 - <<'PY'
import re
cur=None; rows={}
for ln in open("gpurun_out/r04/batch12.txt"):
    if ln.startswith("=="):
        m=re.match(r"== \[(\w+)\] (.*)", ln.strip()); cur=(m.group(2), m.group(1))
    elif ln.startswith("{"):
        m=re.search(r'"temporal_ms_med": ([0-9.]+).*?"sequence_ms_med": ([0-9.]+)', ln)
        if m: rows.setdefault(cur[0],{}).setdefault(cur[1],[]).append((float(m.group(1)),float(m.group(2))))
for k,v in rows.items():
    print(k)
    for lib in ("main","prev"):
        print(f"   {lib}: "+"  ".join(f"{a:.3f} / {b:.3f}" for a,b in v.get(lib,[])))
PY
