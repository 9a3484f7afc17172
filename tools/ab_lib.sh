#!/bin/bash
# Same-box A/B of two builds of libnunif_hip.so (NUNIF_HIP_LIB picks the library, nunif_amd/_hip.py): the swin parity
# tests on the new build first, then the 1080p bench alternating base / new.  Run on the GPU box from the repo root:
#     bash tools/ab_lib.sh <tag> [rounds]
# Writes gpurun_out/<tag>_ab_{base,new}_<i>.json (bench lines) and gpurun_out/<tag>_ab_tests.log.
set -u
TAG=${1:-ab}
ROUNDS=${2:-2}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
BASE=$REPO/nunif_amd/libnunif_hip_base.so
timeout 900 python -m pytest tests/test_gpu_swin.py tests/test_gpu_waifu2x_api.py -m gpu -x -q > "$OUT/${TAG}_ab_tests.log" 2>&1
echo "tests rc=$?" >> "$OUT/${TAG}_ab_tests.log"
tail -5 "$OUT/${TAG}_ab_tests.log"
BENCH="python bench.py --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --steps 60 --warmup 5"
for i in $(seq 1 $ROUNDS); do
    NUNIF_HIP_LIB=$BASE timeout 600 $BENCH > "$OUT/${TAG}_ab_base_$i.json" 2> "$OUT/${TAG}_ab_base_$i.err"
    timeout 600 $BENCH > "$OUT/${TAG}_ab_new_$i.json" 2> "$OUT/${TAG}_ab_new_$i.err"
done
python - <<'PY'
import json, glob, os, sys
out = os.path.join(os.getcwd(), "gpurun_out")
tag = os.environ.get("AB_TAG", "")
for f in sorted(glob.glob(os.path.join(out, "*_ab_*_*.json"))):
    try:
        line = [l for l in open(f) if l.startswith("{")][-1]
        r = json.loads(line)
        ks = {k["name"]: round(k["us_per_launch"], 1) for k in r.get("roofline", {}).get("top_kernels", [])} if isinstance(r.get("roofline"), dict) else {}
        kc = r.get("kernel_classes") or {}
        print(os.path.basename(f), "value", r.get("value"), "single", (r.get("single_stream") or {}).get("value"), "psnr", r.get("psnr_vs_oracle_db"))
        if isinstance(kc, list):
            for k in kc[:12]:
                print("    ", k)
        elif isinstance(kc, dict):
            for k, v in list(kc.items())[:12]:
                print("    ", k, v)
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
