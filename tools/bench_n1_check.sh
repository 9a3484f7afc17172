#!/bin/bash
# VERDICT r05 item 3a, on the GPU box: the N > 1 code path of bench.py on a ONE-rank nccl group (NUNIF_BENCH_FORCE_DIST=1: process
# group, barriers, MAX over ranks, delivery leg, iw3 / cunet / config-5 legs) next to the plain line — `value` must agree within 2 %,
# and `gathered_value` (every frame quantised and handed to rank 0) is printed beside it.
#     bash tools/bench_n1_check.sh <tag>      -> gpurun_out/<tag>_n1_plain.json, gpurun_out/<tag>_n1_dist.json, gpurun_out/<tag>_n1_check.txt
set -u
TAG=${1:-n1}
OUT=$(pwd)/gpurun_out; mkdir -p "$OUT"
COMMON="--steps 60 --warmup 5 --no-cpu-baseline --no-host-frames --no-4k"
timeout 900 python bench.py $COMMON --no-iw3 --no-cunet --no-config5 > "$OUT/${TAG}_n1_plain.json" 2> "$OUT/${TAG}_n1_plain.err"
NUNIF_BENCH_FORCE_DIST=1 timeout 1200 python bench.py --gpus 1 $COMMON > "$OUT/${TAG}_n1_dist.json" 2> "$OUT/${TAG}_n1_dist.err"
python - "$OUT/${TAG}_n1_plain.json" "$OUT/${TAG}_n1_dist.json" <<'PY' | tee "$OUT/${TAG}_n1_check.txt"
import json, sys
def line(p):
    return json.loads([l for l in open(p) if l.startswith("{")][-1])
a, b = line(sys.argv[1]), line(sys.argv[2])
ratio = b["value"] / a["value"]
print(f"plain value {a['value']}  one-rank nccl group value {b['value']}  ratio {ratio:.4f}  gathered_value {b.get('gathered_value')}")
print("multi_gpu", json.dumps(b.get("multi_gpu")))
for k in ("iw3", "cunet", "config5"):
    r = b.get(k) or {}
    print(k, {x: r.get(x) for x in ("value", "fps", "ms_per_frame", "ms_per_frame_per_gpu", "frames", "frames_delivered", "world")})
print("ok", b.get("ok"), "errors", b.get("errors"))
assert abs(ratio - 1.0) <= 0.02, "the N > 1 path at N = 1 disagrees with the plain line by more than 2 %"
assert b.get("ok") and b["gathered"]["frames_delivered"] == b["gathered"]["frames"]
print("PASS")
PY
