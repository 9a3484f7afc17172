#!/bin/bash
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export NUNIF_GEMM_RING=1; else unset NUNIF_GEMM_RING; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --streams 1 > gpurun_out/gr_$v.json 2>/dev/null
  python - <<PY
import json
r=json.loads(open("gpurun_out/gr_$v.json").read().strip().splitlines()[-1])
cs=[c for c in r["kernel_classes"] if c["kernel"].startswith("gemm")]
print("RING=$v  frame MPix/s", r["value"], [(c["kernel"], c["avg_us"], c["launches_per_frame"]) for c in cs])
PY
done
