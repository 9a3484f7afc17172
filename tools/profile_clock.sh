#!/bin/bash
# Effective shader clock per kernel class = GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md "DVFS give-back"): the part
# clocks to its power budget, so a kernel's cycle count and its wall time are two different things.
#     bash tools/profile_clock.sh <tag>   -> gpurun_out/<tag>_clock.txt
set -u
TAG=${1:-clk}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH=${PROF_CMD:-"python $REPO/bench.py --batch-size 45 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --streams 1 --steps 3 --warmup 1"}
rm -rf /tmp/pclk
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pclk -o clk -- $BENCH > "$OUT/${TAG}_clock.log" 2>&1
python - /tmp/pclk > "$OUT/${TAG}_clock.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
ct = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not ct:
    print("no counter csv"); sys.exit(0)
rows = list(csv.DictReader(open(ct[0])))
print("# columns:", list(rows[0].keys()))
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set); t = defaultdict(float); seen = set()
for r in rows:
    name = r["Kernel_Name"][:56]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    did = r.get("Dispatch_Id")
    n[name].add(did)
    if (name, did) not in seen:
        seen.add((name, did))
        if "Start_Timestamp" in r and r.get("Start_Timestamp"):
            t[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        elif did in dur:
            t[name] += dur[did]
print(f"{'kernel':56s} {'launches':>8s} {'us/launch':>10s} {'GUI_ACTIVE/launch':>18s} {'eff GHz':>8s} {'WAVE_CYC/launch':>16s}")
for name in sorted(acc, key=lambda k: -t[k]):
    k = max(1, len(n[name]))
    ga = acc[name].get("GRBM_GUI_ACTIVE", 0.0)
    ghz = ga / t[name] / 1e9 if t[name] else float("nan")
    print(f"{name:56s} {k:8d} {1e6 * t[name] / k:10.1f} {ga / k:18.0f} {ghz:8.3f} {acc[name].get('SQ_WAVE_CYCLES', 0.0) / k:16.0f}")
PY
head -30 "$OUT/${TAG}_clock.txt"
