#!/bin/bash
# Memory-path counter passes (TCP / TA / TCC / SPI; own runs, never with trace domains):
#   PROF_CMD='python tools/da_probe.py vits' bash tools/profile_mem.sh <tag> [env assignments...]
set -u
TAG=${1:-mem}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMD=${PROF_CMD:-"python $REPO/tools/da_probe.py vits"}
: > "$OUT/${TAG}_mem.txt"
i=0
# Round 3: passes with 7-8 TCP / TA / TCC counters each fail with "Request exceeds the capabilities of the hardware to collect" and
# the aborted rocprofv3 then sits until the timeout (4 x 240 s of GPU time lost) — at most 4 counters of one block per pass, and
# a short timeout.  Only the SPI / GRBM pass of the original grouping produced data (profiles/r03_da_spi.txt).
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY SPI_CSN_WINDOW_VALID GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1)); d=/tmp/pmem_$i; rm -rf $d
    timeout ${PROF_TIMEOUT:-60} rocprofv3 --pmc $grp --output-format csv -d $d -o pmc -- $CMD > "$OUT/${TAG}_mem_run$i.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then for c in $grp; do echo "== $c" >> "$OUT/${TAG}_mem.txt"; python $REPO/tools/aggregate_pmc.py "$f" $c | head -${PROF_TOP:-8} >> "$OUT/${TAG}_mem.txt"; done; else echo "pass $i failed" >> "$OUT/${TAG}_mem.txt"; tail -5 "$OUT/${TAG}_mem_run$i.log" >> "$OUT/${TAG}_mem.txt"; fi
done
