#!/bin/bash
# timing-only ablations of the weight-stationary C = 192 tail (needs a NUNIF_BUILD_ABL=1 build; results are wrong)
mkdir -p gpurun_out
for v in ${ABLS:-0 1 2 3 4 7 8 12}; do
  NUNIF_TAIL_WS_ABL=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-frames --streams 1 > gpurun_out/abl_$v.json 2> gpurun_out/abl_$v.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/abl_$v.json").read().strip().splitlines()[-1])
c=[c for c in r["kernel_classes"] if c["kernel"].startswith("proj_mlp_ws")][0]
print("ABL=$v  frame MPix/s", r["value"], " ws tail avg_us", c["avg_us"])
PY
done
