#!/bin/bash
# round 4: the closing run on the committed build — full GPU suite, fuzzers, evidence (counter passes, bench line, rocprofv3 stats)
mkdir -p gpurun_out/r04
timeout -k 10 400 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gputests_final.log 2>&1; tail -2 gpurun_out/r04/gputests_final.log
FUZZ_LO=3071 FUZZ_HI=3271 timeout -k 10 300 python scripts/fuzz_region_fused.py > gpurun_out/r04/fuzz_rf_final.log 2>&1; tail -1 gpurun_out/r04/fuzz_rf_final.log
FUZZ_LO=6253 FUZZ_HI=6453 timeout -k 10 300 python scripts/fuzz_more.py > gpurun_out/r04/fuzz_more_final.log 2>&1; tail -1 gpurun_out/r04/fuzz_more_final.log
timeout -k 10 560 bash scripts/r04_bench_profiles.sh final > gpurun_out/r04/bench_profiles_final.log 2>&1; tail -4 gpurun_out/r04/bench_profiles_final.log | cut -c1-160
