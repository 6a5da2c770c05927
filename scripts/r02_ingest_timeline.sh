#!/bin/bash
# GPU timeline (copies + kernels) of one store -> HBM read with the chunk decode on the GPU: rocprofv3 traces of
# scripts/r02_gpu_decode_ratio.py on the noisy field, summarised by scripts/r02_ingest_timeline.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02; mkdir -p $o
export YEARS=${YEARS:-1} FIELDS=noisy AGGFLY_HIP_INGEST_TRACE=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/rp_timeline -o t -- python3 scripts/r02_gpu_decode_ratio.py > $o/timeline_run.log 2>&1
grep -E "ingest trace|noisy" $o/timeline_run.log | tail -12
python3 scripts/r02_ingest_timeline.py $o/rp_timeline
