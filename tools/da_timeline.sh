#!/bin/bash
# Per-dispatch durations of ONE depth forward (ViT-S, 4 x 1080p) in launch order, gaps included: gpurun_out/da_timeline.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dt
rocprofv3 --kernel-trace --output-format csv -d /tmp/dt -o t -- python $REPO/tools/da_probe.py ${1:-vits} > "$OUT/da_timeline.log" 2>&1
f=$(find /tmp/dt -name '*kernel_trace.csv' | head -1)
python - "$f" > "$OUT/da_timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one forward = from one da_im2col_kernel to the next; take the last complete one
idx = [i for i, r in enumerate(rows) if "da_im2col" in r["Kernel_Name"]]
a, b = idx[-3], idx[-1]
prev_end = None
tot = gaps = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    tot += (e - s) / 1e3; gaps += max(gap, 0.0)
    name = r["Kernel_Name"].replace("nunif::", "").replace("void ", "")[:64]
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>5}  {name}")
print(f"kernels {b - a}  busy {tot:.1f} us  gaps {gaps:.1f} us")
PY
tail -${2:-90} "$OUT/da_timeline.txt"
