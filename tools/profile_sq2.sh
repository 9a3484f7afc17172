#!/bin/bash
# Second set of SQ counter passes (co-execution, memory / LDS latency levels, LDS conflicts, instruction fetch):
#     bash tools/profile_sq2.sh <tag>   -> gpurun_out/<tag>_sq2.txt   (own runs, never with trace domains)
set -u
TAG=${1:-sq2}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --batch-size 45 --no-cpu-baseline --no-host-frames --streams 1 --steps 2 --warmup 1"
: > "$OUT/${TAG}_sq2.txt"
i=0
for grp in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F32"; do
    i=$((i+1)); d=/tmp/psq2_$i; rm -rf $d
    rocprofv3 --pmc $grp --output-format csv -d $d -o pmc -- $BENCH > "$OUT/${TAG}_sq2_run$i.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then for c in $grp; do echo "== $c" >> "$OUT/${TAG}_sq2.txt"; python $REPO/tools/aggregate_pmc.py "$f" $c | head -5 >> "$OUT/${TAG}_sq2.txt"; done; else echo "pass $i failed" >> "$OUT/${TAG}_sq2.txt"; tail -5 "$OUT/${TAG}_sq2_run$i.log" >> "$OUT/${TAG}_sq2.txt"; fi
done
