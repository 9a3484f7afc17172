#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU box from the repo root:
#     bash tools/profile_round.sh r01b
# Writes gpurun_out/<tag>_kernel_stats_bench_b45.csv, gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.txt; copy those into
# profiles/ afterwards.  Counters are collected in their own passes (never together with trace domains).
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --streams 1: per-kernel durations must not be overlapped with a second frame (bench.py measures its roofline leg the same way)
BENCH="python $REPO/bench.py --batch-size 45 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --streams 1"

rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o ks -- $BENCH --steps 6 --warmup 2 > "$OUT/${TAG}_prof_bench.log" 2>&1
f=$(find /tmp/p1 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats_bench_b45.csv"

for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/p_$ctr
    rm -rf $d
    rocprofv3 --pmc $ctr --output-format csv -d $d -o pmc -- $BENCH --steps 2 --warmup 1 >> "$OUT/${TAG}_prof_bench.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $REPO/tools/aggregate_pmc.py "$f" $ctr > "$OUT/${TAG}_pmc_${ctr}.txt"
done
ls -la "$OUT" | grep "$TAG"
