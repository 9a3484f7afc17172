#!/bin/bash
# Same-box comparison of several builds of libnunif_hip.so (tools/build_variant.py): the swin parity tests on every variant, then
# the 1080p bench cycling base -> variants, ROUNDS times.  Run on the GPU box from the repo root:
#     bash tools/ab_multi.sh <tag> <rounds> <variant> [<variant> ...]        (variant "base" = nunif_amd/libnunif_hip_base.so)
# Writes gpurun_out/<tag>_<variant>_<i>.json (bench lines), gpurun_out/<tag>_<variant>_tests.log, gpurun_out/<tag>_summary.txt.
set -u
TAG=$1; ROUNDS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
TESTS=${AB_TESTS:-"tests/test_gpu_swin.py tests/test_gpu_waifu2x_api.py"}
BENCH=${AB_BENCH:-"python bench.py --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --steps 60 --warmup 5"}
for v in "$@"; do
    [ "$v" = base ] && continue
    NUNIF_HIP_LIB=$REPO/nunif_amd/libnunif_hip_$v.so timeout 900 python -m pytest $TESTS -m gpu -x -q > "$OUT/${TAG}_${v}_tests.log" 2>&1
    echo "$v tests rc=$? $(tail -1 "$OUT/${TAG}_${v}_tests.log")" | tee -a "$OUT/${TAG}_summary.txt"
done
for i in $(seq 1 $ROUNDS); do
    for v in "$@"; do
        NUNIF_HIP_LIB=$REPO/nunif_amd/libnunif_hip_$v.so timeout 600 $BENCH > "$OUT/${TAG}_${v}_$i.json" 2> "$OUT/${TAG}_${v}_$i.err"
    done
done
AB_TAG=$TAG python - "$@" <<'PY' | tee -a "$OUT/${TAG}_summary.txt"
import json, os, sys, glob
out = os.path.join(os.getcwd(), "gpurun_out"); tag = os.environ["AB_TAG"]
rows = {}
for v in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(out, f"{tag}_{v}_[0-9]*.json"))):
        try:
            r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        except Exception as e:
            print(os.path.basename(f), "unreadable", e); continue
        kc = r.get("kernel_classes") or {}
        ks = {}
        if isinstance(kc, dict):
            for k, x in kc.items():
                ks[k] = x.get("us_per_launch") if isinstance(x, dict) else x
        elif isinstance(kc, list):
            for x in kc:
                if isinstance(x, dict): ks[x.get("kernel")] = x.get("avg_us")
        rows.setdefault(v, []).append((r.get("value"), (r.get("single_stream") or {}).get("value"), r.get("psnr_vs_oracle_db"), ks))
names = []
for v, rs in rows.items():
    for r in rs:
        for k in r[3]:
            if k not in names: names.append(k)
for v, rs in rows.items():
    print(f"{v:>10}: value " + " ".join(f"{r[0]:.1f}" for r in rs) + "  single " + " ".join(f"{r[1]:.1f}" if r[1] else "-" for r in rs) + "  psnr " + " ".join(str(r[2]) for r in rs))
print("kernel us_per_launch (mean over rounds):")
for k in names[:14]:
    line = f"  {str(k)[:58]:<58}"
    for v, rs in rows.items():
        xs = [r[3].get(k) for r in rs if r[3].get(k) is not None]
        line += f" {v}={sum(xs)/len(xs):8.1f}" if xs else f" {v}=   -"
    print(line)
PY
