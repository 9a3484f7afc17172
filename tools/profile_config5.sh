#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of bench.py's config5 record.
#     bash tools/profile_config5.sh <tag>   -> gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.txt
set -u
TAG=${1:-r04f}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/p5_$ctr; rm -rf $d
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $d -o pmc -- python $REPO/tools/config5_probe.py > "$OUT/${TAG}_pmc_$ctr.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $REPO/tools/aggregate_pmc.py "$f" $ctr > "$OUT/${TAG}_pmc_${ctr}.txt"
done
ls -la "$OUT" | grep "${TAG}_pmc"
