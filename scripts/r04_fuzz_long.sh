#!/bin/bash
# round 4: long fuzz runs on the closing build (numbers against the oracle; every seed prints only on failure, progress every 100 seeds)
mkdir -p gpurun_out/r04
FUZZ_LO=3447 FUZZ_HI=4247 timeout -k 10 420 python scripts/fuzz_more.py > gpurun_out/r04/fuzz_more_long.log 2>&1; tail -1 gpurun_out/r04/fuzz_more_long.log
FUZZ_LO=1064 FUZZ_HI=1464 timeout -k 10 300 python scripts/fuzz_region_fused.py > gpurun_out/r04/fuzz_rf_long.log 2>&1; tail -1 gpurun_out/r04/fuzz_rf_long.log
FUZZ_LO=240 FUZZ_HI=640 timeout -k 10 240 python scripts/fuzz_ingest.py > gpurun_out/r04/fuzz_ingest_long.log 2>&1; tail -1 gpurun_out/r04/fuzz_ingest_long.log
