#!/bin/bash
# A/B, alternating processes: form 8 of the period end (DPP moves) against form 6 (ds_bpermute; scripts/probe/_build/libaggfly_hip_prev.so = 3f700a0e616b), more shapes
mkdir -p gpurun_out/r04
FUZZ_LO=1465 FUZZ_HI=1765 timeout -k 10 300 python scripts/fuzz_region_fused.py > gpurun_out/r04/fuzz_rf_dpp.log 2>&1; tail -1 gpurun_out/r04/fuzz_rf_dpp.log
out=gpurun_out/r04/batch9.txt
: > $out
run() { echo "== [$LIBTAG] $*" >> $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{' | cut -c1-330 >> $out; }
for rep in 1 2 3; do
for lib in main prev; do
  export LIBTAG=$lib
  if [ $lib = prev ]; then export AGGFLY_HIP_LIB=$PWD/scripts/probe/_build/libaggfly_hip_prev.so; else unset AGGFLY_HIP_LIB; fi
  run --plan c2 --dtype f64 --periods 365 --arms base
  run --plan c2 --dtype f64 --periods 12 --arms base
  run --plan c5 --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 52 --regions 40000 --arms base
  run --plan meanpoly --dtype f32 --T 730 --ny 1801 --nx 3600 --spd 2 --periods 365 --regions 40000 --arms base
  run --plan meanpoly --dtype f32 --T 1095 --ny 721 --nx 1440 --spd 3 --periods 365 --arms base
  run --plan meanpoly --dtype f64 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 365 --arms base
  run --plan c2 --dtype f32 --periods 73 --arms base
done; done
python3 - <<'PY'
import re
cur=None; rows={}
for ln in open("gpurun_out/r04/batch9.txt"):
    if ln.startswith("=="):
        m=re.match(r"== \[(\w+)\] (.*)", ln.strip()); cur=(m.group(2), m.group(1))
    elif ln.startswith("{"):
        m=re.search(r'"temporal_ms_med": ([0-9.]+).*?"sequence_ms_med": ([0-9.]+)', ln)
        if m: rows.setdefault(cur[0],{}).setdefault(cur[1],[]).append((float(m.group(1)),float(m.group(2))))
for k,v in rows.items():
    print(k)
    for lib,name in (("main","form 8"),("prev","form 6")):
        print(f"   {name}: "+"  ".join(f"{a:.3f} / {b:.3f}" for a,b in v.get(lib,[])))
PY
