#!/bin/bash
# Round 4's evidence run (round 3's r03_bench_profiles2.sh + the configs[3] traffic passes): the counter passes come FIRST and their results are merged into profiles/traffic.json /
# profiles/valu_counts.json on the box before `python bench.py` runs — so the bench line's `traffic` is measured on the build it times
# (`traffic_measured_on_this_build: true`).  usage: scripts/r04_bench_profiles2.sh TAG      (outputs under gpurun_out/r04/)
tag=${1:-final}
export ROUND=r04
o=gpurun_out/r04; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/pmc_bench_${tag}_$c -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $o/pmc_bench_${tag}_$c.log 2>&1 || echo "pass $c failed"
done
export AGGFLY_BENCH_ONLY=C4
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/pmc_bench_${tag}_c4_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > $o/pmc_bench_${tag}_c4_$c.log 2>&1 || echo "pass c4 $c failed"
done
export AGGFLY_BENCH_ONLY=C5
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $o/pmc_bench_${tag}_c5_$n -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > $o/pmc_bench_${tag}_c5_$n.log 2>&1 || echo "pass $n failed"
done
export AGGFLY_BENCH_ONLY=C5_iid
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $o/pmc_bench_${tag}_c5iid_$n -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > $o/pmc_bench_${tag}_c5iid_$n.log 2>&1 || echo "pass iid $n failed"
done
unset AGGFLY_BENCH_ONLY
python3 scripts/pmc_bench_post.py $tag > $o/pmc_post_$tag.log 2>&1
python3 - <<PY
import json
for src, dst in (("$o/traffic_$tag.json", "profiles/traffic.json"), ("$o/valu_counts_$tag.json", "profiles/valu_counts.json")):
    try:
        new = json.load(open(src)); cur = json.load(open(dst)); cur.update(new); json.dump(cur, open(dst, "w"), indent=1)
        json.dump(cur, open("$o/" + dst.split("/")[1].replace(".json", "_merged_$tag.json"), "w"), indent=1)
        print("merged", src, "->", dst)
    except Exception as e:
        print("not merged:", src, e)
PY
python bench.py > $o/bench_$tag.json 2> $o/bench_$tag.err; tail -c 400 $o/bench_$tag.err; head -c 600 $o/bench_$tag.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $o/rp_bench_$tag -o p -- python3 bench.py --steps 20 --no-cpu-baseline --no-other-configs > $o/rp_bench_$tag.log 2>&1
python3 scripts/pmc_bench_post.py $tag > $o/pmc_post2_$tag.log 2>&1; head -12 $o/pmc_post2_$tag.log
