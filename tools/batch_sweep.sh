#!/bin/bash
# per-kernel-class time per frame at several tile batches (single stream): does a working set inside the 256 MB memory-side
# cache make the level-1 (C = 96) kernels faster per token?
mkdir -p gpurun_out
for b in ${BATCHES:-45 15 9 5}; do
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --streams 1 --batch-size $b > gpurun_out/bs_$b.json 2> gpurun_out/bs_$b.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/bs_$b.json").read().strip().splitlines()[-1])
print("batch $b  MPix/s", r["value"], " ms/frame", r["ms_per_step"])
for c in r["kernel_classes"][:8]:
    print("    %-40s launches/frame %4d avg_us %8.1f  per-frame ms %.3f  TF %.0f GB/s %.0f" % (c["kernel"][:40], c["launches_per_frame"], c["avg_us"], c["avg_us"] * c["launches_per_frame"] * 1e-3, c["tflops"], c["gbs"]))
PY
done
