#!/bin/bash
# call 49: one-rank nccl smoke of the N > 1 bench legs incl. the config-5 replicas leg
cd /root/repo
mkdir -p gpurun_out
LEG_FRAMES=8 timeout 100 python tools/bench_legs_smoke.py > gpurun_out/r05fin5_legs_smoke.json 2> gpurun_out/r05fin5_legs_smoke.err
echo "rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05fin5_legs_smoke.json"))
print({k: (v.get("value"), v.get("fps"), v.get("scaling")) for k, v in r.items() if isinstance(v, dict) and "value" in v})
PY
tail -3 gpurun_out/r05fin5_legs_smoke.err
