#!/bin/bash
# rocprofv3 evidence for the HBM-bound iw3 kernels (forward warp, backward warp, dilate_edge, resize, stitch, to_frame):
#     bash tools/profile_iw3_ops.sh r02   -> gpurun_out/<tag>_iw3ops_kernel_stats.csv, <tag>_iw3ops_pmc_{FETCH,WRITE}_SIZE.txt
# Counters are collected in their own passes (never together with trace domains).
set -u
TAG=${1:-rXX}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/bench_ops.py"
rm -rf /tmp/q1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q1 -o ks -- $CMD > "$OUT/${TAG}_iw3ops.log" 2>&1
f=$(find /tmp/q1 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_iw3ops_kernel_stats.csv"
# PMC passes on the exact launch shape bench.py prices (tools/bench_fw.py); written as <tag>i_pmc_* so that bench.py's
# pmc_traffic_bytes() finds the forward-warp kernel there and the swin kernels in <tag>_pmc_*
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/q_$ctr; rm -rf $d
    rocprofv3 --pmc $ctr --output-format csv -d $d -o pmc -- python $REPO/tools/bench_fw.py 10 >> "$OUT/${TAG}_iw3ops.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $REPO/tools/aggregate_pmc.py "$f" $ctr > "$OUT/${TAG}i_pmc_${ctr}.txt"
done
ls -la "$OUT" | grep "${TAG}"
