#!/bin/bash
# timing-only ablations of the fused qkv + window attention kernels (needs a NUNIF_BUILD_ABL=1 build; results are wrong)
mkdir -p gpurun_out
for v in ${ABLS:-0 1 2 3}; do
  NUNIF_ATTN_ABL=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-frames --streams 1 > gpurun_out/attn_abl_$v.json 2> gpurun_out/attn_abl_$v.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/attn_abl_$v.json").read().strip().splitlines()[-1])
c=[(c["kernel"], c["avg_us"]) for c in r["kernel_classes"] if c["kernel"].startswith("qkv_attn")]
print("ATTN_ABL=$v  frame MPix/s", r["value"], c)
PY
done
