#!/bin/bash
# round 4: long fuzz runs on the closing build (numbers against the oracle; every seed prints only on failure, progress every 100 seeds)
mkdir -p gpurun_out/r04
FUZZ_LO=4850 FUZZ_HI=5650 timeout -k 10 420 python scripts/fuzz_more.py > gpurun_out/r04/fuzz_more_long.log 2>&1; tail -1 gpurun_out/r04/fuzz_more_long.log
FUZZ_LO=2168 FUZZ_HI=2568 timeout -k 10 300 python scripts/fuzz_region_fused.py > gpurun_out/r04/fuzz_rf_long.log 2>&1; tail -1 gpurun_out/r04/fuzz_rf_long.log
FUZZ_LO=640 FUZZ_HI=1040 timeout -k 10 240 python scripts/fuzz_ingest.py > gpurun_out/r04/fuzz_ingest_long.log 2>&1; tail -1 gpurun_out/r04/fuzz_ingest_long.log
FUZZ_LO=200 FUZZ_HI=500 timeout -k 10 240 python scripts/fuzz_modes.py > gpurun_out/r04/fuzz_modes_long.log 2>&1; tail -1 gpurun_out/r04/fuzz_modes_long.log
