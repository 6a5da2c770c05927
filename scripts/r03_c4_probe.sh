#!/bin/bash
# Round 3, C4 (13 bins, LDS histogram): which unit bounds it?  knob sweep in one process + PMC passes.  scripts/r03_c4_probe.sh TAG
tag=${1:-a}
o=gpurun_out/r03; mkdir -p $o
S="--plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600"
python scripts/r03_arms.py $S --data era5 --rounds 7 --out $o/c4_arms_$tag.json --arms base \
   "AFHIP_WGS_PER_CU=2" "AFHIP_WGS_PER_CU=6" "AFHIP_WGS_PER_CU=8" "AFHIP_WGS_PER_CU=12" "AFHIP_WGS_PER_CU=16" "AFHIP_WGS_PER_CU=24" \
   "AFHIP_FORCE_WG=64" "AFHIP_FORCE_WG=64,AFHIP_WGS_PER_CU=8" "AFHIP_FORCE_WG=128" "AFHIP_FORCE_WG=128,AFHIP_WGS_PER_CU=8" \
   "AFHIP_NO_ARITH_EDGES=1" "AFHIP_NO_ARITH_EDGES=1,AFHIP_WGS_PER_CU=8" "tuning=208" "tuning=208,AFHIP_WGS_PER_CU=8" \
   "AFHIP_XCD_REMAP=0" > $o/c4_arms_$tag.log 2>&1 && grep -E '^\{' $o/c4_arms_$tag.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "MeanOccupancyPerCU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_r03c4${tag}_$i -o p -- python3 scripts/kbench.py $S --data era5 --tunings 0 --rounds 2 > gpurun_out/pmc_r03c4${tag}_$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -2 gpurun_out/pmc_r03c4${tag}_$i.log; }
done
python3 scripts/pmc_merge.py r03c4${tag} > $o/pmc_c4_$tag.txt 2>&1; tail -40 $o/pmc_c4_$tag.txt
