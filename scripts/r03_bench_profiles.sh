#!/bin/bash
# Evidence for the bench line, all from bench.py itself (run on the GPU box):
#   1. python bench.py                                            -> gpurun_out/r03/bench_<tag>.json
#   2. rocprofv3 --kernel-trace --stats of the same command       -> gpurun_out/r03/bench_<tag>_kernel_stats.csv
#   3. --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs)       -> profiles-ready traffic entry (scripts/pmc_bench_post.py)
#   4. --pmc SQ_INSTS_VALU ... over the C5 config only            -> VALU instructions per cell-step of the sine_dd kernel
# usage: scripts/r03_bench_profiles.sh TAG
tag=${1:-final}
export ROUND=r03
o=gpurun_out/r03; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $o/bench_$tag.json 2> $o/bench_$tag.err; tail -c 600 $o/bench_$tag.err; head -c 1200 $o/bench_$tag.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $o/rp_bench_$tag -o p -- python3 bench.py --steps 20 --no-cpu-baseline --no-other-configs > $o/rp_bench_$tag.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/pmc_bench_${tag}_$c -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $o/pmc_bench_${tag}_$c.log 2>&1 || echo "pass $c failed"
done
export AGGFLY_BENCH_ONLY=C5
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $o/pmc_bench_${tag}_c5_$n -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $o/pmc_bench_${tag}_c5_$n.log 2>&1 || echo "pass $n failed"
done
unset AGGFLY_BENCH_ONLY
python3 scripts/pmc_bench_post.py $tag
