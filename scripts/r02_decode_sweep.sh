#!/bin/bash
# Batch size x staging slots of the GPU chunk decode route, 1-year and 4-year stores, noisy and smooth fields.
o=gpurun_out/r02; mkdir -p $o
for years in 1 4; do for f in noisy "smooth, 0.01"; do for mb in 128 256 512; do for slots in 2 4; do
  echo "== years $years field $f batch $mb MB slots $slots"
  YEARS=$years FIELDS="$f" AGGFLY_HIP_GPU_DECODE_BATCH_MB=$mb AGGFLY_HIP_GPU_DECODE_SLOTS=$slots timeout -k 10 200 python3 scripts/r02_gpu_decode_ratio.py 2>&1 | grep gpu_decode_GBps
done; done; done; done
