#!/bin/bash
# Per-dispatch kernel durations of ONE frame (single stream), in launch order: which launches of a kernel class are slow.
#     bash tools/kernel_timeline.sh   -> gpurun_out/timeline.txt
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o kt -- python $REPO/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k > "$OUT/timeline.log" 2>&1
f=$(find /tmp/pt -name '*kernel_trace.csv' | head -1)
python - "$f" > "$OUT/timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full frame: find the last stitch kernel and walk back to the previous one
idx = [i for i, r in enumerate(rows) if "stitch" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    name = r["Kernel_Name"].replace("nunif::", "").replace("void ", "")[:60]
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {name}")
PY
tail -50 "$OUT/timeline.txt"
