#!/bin/bash
# PMC passes over one kbench workload (counters in their own runs, kernel-trace only).
#   scripts/pmc_pass.sh <tag> <kbench args...>
# Writes gpurun_out/pmc_<tag>_<set>/ and a merged summary gpurun_out/pmc_<tag>.csv
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python3 scripts/kbench.py "$@" --rounds 2 > gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python3 scripts/pmc_merge.py $tag
