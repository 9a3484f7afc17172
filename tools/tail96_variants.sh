#!/bin/bash
for v in 6 1 2 3 7 6; do
  NUNIF_TAIL_VARIANT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --streams 1 > gpurun_out/tv_$v.json 2>/dev/null
  python - <<PY
import json
r=json.loads(open("gpurun_out/tv_$v.json").read().strip().splitlines()[-1])
c=[c for c in r["kernel_classes"] if c["kernel"].startswith("proj_mlp_r")][0]
print("variant $v  frame MPix/s", r["value"], " tail96", c["kernel"], c["avg_us"])
PY
done
