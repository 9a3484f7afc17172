#!/bin/bash
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o t -- python $REPO/tools/da_probe.py vits > /dev/null 2>&1
f=$(find /tmp/ct -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# last forward only: take the last 200 dispatches
sel = [r for r in rows if "conv_kernel" in r["Kernel_Name"] or "gemm_os" in r["Kernel_Name"] or "da_attn" in r["Kernel_Name"]]
agg = collections.OrderedDict()
for r in rows[-260:]:
    k = r["Kernel_Name"][:60]
    if "conv" not in k: continue
    key = (k, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Workgroup_Size_X", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    print(k, len(v), "avg_us %.1f" % (sum(v) / len(v)))
PY
