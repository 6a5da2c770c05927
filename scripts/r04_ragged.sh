#!/bin/bash
# round 4: inner groups of mixed lengths one to four rows (a sub-daily series with gaps): the `_rag` form against the general path
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -k "mixed_short or gaps_f32 or 47 or 53 or 59 or 65" > gpurun_out/r04/ragged_tests.log 2>&1; tail -3 gpurun_out/r04/ragged_tests.log
o=gpurun_out/r04/ragged_arms_raw.txt; : > $o
run() { echo "== $*" >> $o; timeout -k 10 200 python scripts/r03_arms.py "$@" >> $o 2>&1; }
run --plan meanpoly --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --gaps 0.02 --periods 1 --regions 40000 --arms base AFHIP_NO_RAGGED_MODE=1
run --plan meanpoly --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --gaps 0.5 --periods 1 --regions 40000 --arms base AFHIP_NO_RAGGED_MODE=1
run --plan meanpoly --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --periods 1 --regions 40000 --arms base
run --plan mean --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --gaps 0.02 --periods 1 --regions 40000 --arms base AFHIP_NO_RAGGED_MODE=1
run --plan c1 --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --gaps 0.02 --periods 1 --regions 40000 --arms base AFHIP_NO_RAGGED_MODE=1
run --plan meanpoly --dtype f64 --T 1460 --ny 721 --nx 1440 --spd 4 --gaps 0.02 --periods 1 --arms base AFHIP_NO_RAGGED_MODE=1
run --plan mean --dtype f64 --T 1460 --ny 721 --nx 1440 --spd 4 --gaps 0.02 --periods 1 --arms base AFHIP_NO_RAGGED_MODE=1
run --plan meanpoly --dtype f32 --T 1460 --ny 721 --nx 1440 --spd 4 --gaps 0.02 --periods 12 --arms base AFHIP_NO_REGION_FUSED=1 AFHIP_NO_RAGGED_MODE=1
run --plan c5 --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --gaps 0.02 --periods 1 --regions 40000 --arms base AFHIP_NO_RAGGED_MODE=1
python scripts/r04_fmt_arms.py $o > gpurun_out/r04/ragged_arms.txt; cat gpurun_out/r04/ragged_arms.txt
