#!/bin/bash
# tests + A/B of the C = 192 swin tail (weight-stationary kernel vs the round-1 LDS-ring kernel) + ablations, same box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_swin.py -m gpu -x -q -k "stagewise or golden or batch_invariance or 1080p" > gpurun_out/ws_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/ws_tests.log
tail -4 gpurun_out/ws_tests.log
for v in 1 0; do
  NUNIF_TAIL_WS=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-frames > gpurun_out/ws_bench_$v.json 2> gpurun_out/ws_bench_$v.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/ws_bench_$v.json").read().strip().splitlines()[-1])
print("TAIL_WS=$v value", r["value"], "single", r.get("single_stream"))
for c in r["kernel_classes"][:5]: print("   ", c)
PY
done
ABLS="${ABLS:-0 1 2 3 8}" tools/abl_tail_ws.sh
