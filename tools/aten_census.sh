#!/bin/bash
# Which ATen (at::native::*) kernels still run in the iw3 frame loop?  rocprofv3 --kernel-trace of tools/bench_iw3_sched.py.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o ks -- python $REPO/tools/bench_iw3_sched.py --frames 32 --batch 4 --methods ${METHODS:-row_flow_v3,forward_fill} > "$OUT/aten_census.log" 2>&1
f=$(find /tmp/pa -name '*kernel_stats.csv' | head -1)
cp "$f" "$OUT/iw3_sched_kernel_stats.csv"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
aten = [r for r in rows if "at::native" in r["Name"] or "at_cuda" in r["Name"] or "elementwise" in r["Name"].lower()]
print("total kernel time ms", tot / 1e6, " ATen share %.2f %%" % (100 * sum(float(r["TotalDurationNs"]) for r in aten) / tot))
for r in sorted(aten, key=lambda r: -float(r["TotalDurationNs"])):
    print(f'{int(r["Calls"]):6d} calls {float(r["TotalDurationNs"])/1e3:10.1f} us  {r["Name"][:150]}')
PY
