#!/bin/bash
# Spatial-stage kernels on near-uniform (k-d split) and log-normal (four decades of row lengths) weights tables:
# the table-order serial kernel (AFHIP_SPMM_SERIAL=1) against the default (wave per row segment / segments + combine).
#   scripts/r02_spmm_skew.sh  -> gpurun_out/r02/spmm_skew.txt   (rocprofv3 --kernel-trace --stats, engine kernels only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02; mkdir -p $o
run() {
  tag=$1; serial=$2; shift 2
  if [ "$serial" = 1 ]; then export AFHIP_SPMM_SERIAL=1; else unset AFHIP_SPMM_SERIAL; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/rp_$tag -o p -- python3 scripts/kbench.py "$@" --tunings 0 --rounds 4 > $o/rp_$tag.log 2>&1
  python3 - "$tag" "$o" <<'PY'
import sys, glob, pandas as pd
tag, o = sys.argv[1], sys.argv[2]
d = pd.read_csv(glob.glob(f"{o}/rp_{tag}/**/*kernel_stats.csv", recursive=True)[0])
d = d[d.Name.str.contains("afhip") & ~d.Name.str.contains("k_fused_temporal")].copy()
d["Name"] = d["Name"].str.replace(r"\(.*", "", regex=True).str.replace("void afhip::", "").str.replace("afhip::", "")
d["avg_us"] = d["AverageNs"] / 1e3
print(f"== {tag}: " + open(f"{o}/rp_{tag}.log").read().split("weights table: ")[1].splitlines()[0])
print(d[["Name", "Calls", "avg_us"]].to_string(index=False))
PY
}
C5="--plan c5 --dtype f32 --ny 1801 --nx 3600 --T 730 --spd 2 --regions 40000"
C4="--plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600"
C2="--plan c2 --dtype f64"
{
run c5_uniform_serial 1 $C5
run c5_uniform_default 0 $C5
run c5_lognormal_serial 1 $C5 --skew lognormal
run c5_lognormal_default 0 $C5 --skew lognormal
run c4_uniform_serial 1 $C4
run c4_uniform_default 0 $C4
run c4_lognormal_serial 1 $C4 --skew lognormal
run c4_lognormal_default 0 $C4 --skew lognormal
run c2_uniform_serial 1 $C2
run c2_uniform_default 0 $C2
run c2_lognormal_serial 1 $C2 --skew lognormal
run c2_lognormal_default 0 $C2 --skew lognormal
} > $o/spmm_skew.txt 2>&1
cat $o/spmm_skew.txt
