#!/bin/bash
# Kernel stats + HBM traffic (FETCH_SIZE / WRITE_SIZE, own passes) of BASELINE configs[2] on one GPU (swin_unet 4x, 4K frame).
#     bash tools/profile_4k.sh <tag>   -> gpurun_out/<tag>_kernel_stats_4k.csv, gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.txt, <tag>_4k_record.json
set -u
TAG=${1:-r06k}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
SCALE4X_PROF=1 timeout 300 python tools/scale4x_probe.py > "$OUT/${TAG}_4k_record.json" 2> "$OUT/${TAG}_4k_record.err"
CMD="python $REPO/tools/scale4x_probe.py"
cd /tmp && export TMPDIR=/tmp
d=/tmp/p4k_stats; rm -rf $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- $CMD > "$OUT/${TAG}_4k_stats.log" 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats_4k.csv"
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/p4k_$ctr; rm -rf $d
    SCALE4X_ITERS=1 timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $d -o pmc -- $CMD > "$OUT/${TAG}_pmc_$ctr.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $REPO/tools/aggregate_pmc.py "$f" $ctr > "$OUT/${TAG}_pmc_${ctr}.txt"
done
head -12 "$OUT/${TAG}_kernel_stats_4k.csv" 2>/dev/null | cut -c1-160
