#!/bin/bash
# A/B of two library builds on one box, alternating processes: the tree against the previous build (scripts/probe/_build/libaggfly_hip_prev.so = 3f700a0e616b);
# first used for form 7 of the period end (fewer VALU: level), then for form 8 (the scan moves its values by DPP instead of ds_bpermute)
mkdir -p gpurun_out/r04
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -k "region_fused or rf or many_period or launch_shape" > gpurun_out/r04/batch8_tests.log 2>&1; tail -2 gpurun_out/r04/batch8_tests.log
out=gpurun_out/r04/batch8.txt
: > $out
run() { echo "== [$LIBTAG] $*" >> $out; timeout -k 10 300 python scripts/r03_arms.py "$@" 2>&1 | grep -E '^\{' | cut -c1-330 >> $out; }
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
  run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --periods 365 --arms base
done; done
python3 - <<'PY'
import json, re
cur = None
rows = {}
for ln in open("gpurun_out/r04/batch8.txt"):
    if ln.startswith("=="):
        m = re.match(r"== \[(\w+)\] (.*)", ln.strip()); cur = (m.group(2), m.group(1))
    elif ln.startswith("{"):
        try: d = json.loads(ln)
        except ValueError:
            m = re.search(r'"temporal_ms_med": ([0-9.]+).*?"sequence_ms_med": ([0-9.]+)', ln); d = {"temporal_ms_med": float(m.group(1)), "sequence_ms_med": float(m.group(2))} if m else None
        if d: rows.setdefault(cur[0], {}).setdefault(cur[1], []).append((d["temporal_ms_med"], d["sequence_ms_med"]))
for k, v in rows.items():
    print(k)
    for lib in ("main", "prev"):
        print(f"   {lib}: " + "  ".join(f"temporal {a:.3f} / step {b:.3f}" for a, b in v.get(lib, [])))
PY
