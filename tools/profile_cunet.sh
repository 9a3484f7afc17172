#!/bin/bash
# SQ counters + HBM traffic (FETCH_SIZE / WRITE_SIZE) of the cunet 1080p render (whole-frame tile batch).
#     bash tools/profile_cunet.sh <tag>   -> gpurun_out/<tag>_sq.txt, gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.txt
set -u
TAG=${1:-r04c}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export PROF_CMD="env CUNET_BATCH=66 CUNET_ONLY=cunet CUNET_ITERS=3 python $REPO/tools/cunet_probe.py"
PROF_TOP=12 bash tools/profile_sq.sh $TAG
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pc_$ctr; rm -rf $d
    timeout 120 rocprofv3 --pmc $ctr --output-format csv -d $d -o pmc -- $PROF_CMD > "$OUT/${TAG}_pmc_$ctr.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $REPO/tools/aggregate_pmc.py "$f" $ctr > "$OUT/${TAG}_pmc_${ctr}.txt"
done
