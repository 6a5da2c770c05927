#!/bin/bash
# C5 sine_dd kernel check: parity tests, accuracy, kbench (iid / era5-like / pair mode off).  scripts/r02_c5.sh TAG
tag=${1:-v}
o=gpurun_out/r02; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q > $o/gputest_$tag.log 2>&1; echo "pytest rc=$?" >> $o/gputest_$tag.log; tail -3 $o/gputest_$tag.log
python scripts/sine_accuracy.py > $o/sine_accuracy_$tag.json 2>&1; cat $o/sine_accuracy_$tag.json
K="python scripts/kbench.py --plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --rounds 5"
$K --tunings 0,108 > $o/kbench_c5_$tag.log 2>&1; grep -E "^tuning|temporal" $o/kbench_c5_$tag.log
AFHIP_NO_PAIR_MODE=1 $K --tunings 0 > $o/kbench_c5_${tag}_nopair.log 2>&1; grep -E "temporal" $o/kbench_c5_${tag}_nopair.log
$K --tunings 0 --data era5 > $o/kbench_c5_${tag}_era5.log 2>&1; grep -E "temporal" $o/kbench_c5_${tag}_era5.log
