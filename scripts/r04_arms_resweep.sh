#!/bin/bash
# round 4: the load-path arms re-measured under single-wave workgroups (round 3 changed the workgroup size after round 2's arm sweep) on the f32 light plans
# needs the arms build: make -C aggfly_amd/csrc MENU=arms BUILD=_build_arms OUT=../../scripts/probe/_build/libaggfly_hip_arms.so
mkdir -p gpurun_out/r04
out=gpurun_out/r04/arms_resweep.txt
: > $out
export AGGFLY_HIP_LIB=$PWD/scripts/probe/_build/libaggfly_hip_arms.so
run() { echo "== $*" | tee -a $out; timeout -k 10 400 python scripts/kbench.py "$@" 2>&1 | grep -vE "amdgpu.ids|^weights table" | cut -c1-260 | tee -a $out; }
run --plan c1 --dtype f32 --data era5 --tunings 0,104,108,204,208,404,408
run --plan c1 --dtype f32 --data era5 --T 350640 --ny 104 --nx 236 --periods 40 --tunings 0,104,108,204,208,404
run --plan c2 --dtype f32 --data era5 --tunings 0,104,108,204,208
run --plan c2 --dtype f64 --data era5 --tunings 0,104,108,204,208
