#!/bin/bash
# PMC passes over the C5 sine_dd kernel: VALU instruction count and issue utilisation.  scripts/r02_pmc_c5.sh TAG [kbench args]
tag=${1:-c5}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
K="scripts/kbench.py --plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000 --tunings 0 --rounds 2 $@"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python3 $K > gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${tag}_$i.log; }
done
python3 scripts/pmc_merge.py $tag > gpurun_out/r02/pmc_$tag.txt 2>&1; cat gpurun_out/r02/pmc_$tag.txt | tail -30
