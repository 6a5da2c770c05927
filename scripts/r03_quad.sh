#!/bin/bash
# four-row groups (6-hourly data): parity first, then lean four-row form vs the general ring path, same process, interleaved
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "four_row or pair_plans" > gpurun_out/r03/quad_tests.log 2>&1 || { tail -40 gpurun_out/r03/quad_tests.log; exit 1; }
tail -3 gpurun_out/r03/quad_tests.log
python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu > gpurun_out/r03/quad_fuzz.log 2>&1 || { tail -40 gpurun_out/r03/quad_fuzz.log; exit 1; }
tail -3 gpurun_out/r03/quad_fuzz.log
for dt in f32 f64; do
  for plan in meanpoly c1 mean; do
    python scripts/r03_arms.py --plan $plan --dtype $dt --T 1460 --ny 721 --nx 1440 --spd 4 --periods 1 --rounds 9 \
        --arms base AFHIP_NO_QUAD_MODE=1 AFHIP_LEAN_STAT1_MIN_K=1 --out gpurun_out/r03/quad_${plan}_${dt}.json > gpurun_out/r03/quad_${plan}_${dt}.log 2>&1
    grep temporal_ms_med gpurun_out/r03/quad_${plan}_${dt}.log
  done
done
python scripts/r03_arms.py --plan meanpoly --dtype f32 --T 1460 --ny 1801 --nx 3600 --spd 4 --periods 1 --rounds 7 \
    --arms base AFHIP_NO_QUAD_MODE=1 tuning=204 --out gpurun_out/r03/quad_meanpoly_f32_big.json > gpurun_out/r03/quad_meanpoly_f32_big.log 2>&1
grep temporal_ms_med gpurun_out/r03/quad_meanpoly_f32_big.log
