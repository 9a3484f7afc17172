#!/bin/bash
# SQ counter passes (own runs, never with trace domains): bash tools/profile_sq.sh <tag> [env assignments...]
# PROF_CMD='python tools/da_probe.py vits' profiles another command (paths relative to the repo root are resolved before the cd)
set -u
TAG=${1:-sq}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
BENCH=${PROF_CMD:-"python $REPO/bench.py --batch-size 45 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --streams 1 --steps 2 --warmup 1"}
: > "$OUT/${TAG}_sq.txt"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM"; do
    i=$((i+1)); d=/tmp/psq_$i; rm -rf $d
    rocprofv3 --pmc $grp --output-format csv -d $d -o pmc -- $BENCH > "$OUT/${TAG}_sq_run$i.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then for c in $grp; do echo "== $c" >> "$OUT/${TAG}_sq.txt"; python $REPO/tools/aggregate_pmc.py "$f" $c | head -${PROF_TOP:-6} >> "$OUT/${TAG}_sq.txt"; done; else echo "pass $i failed" >> "$OUT/${TAG}_sq.txt"; tail -5 "$OUT/${TAG}_sq_run$i.log" >> "$OUT/${TAG}_sq.txt"; fi
done
