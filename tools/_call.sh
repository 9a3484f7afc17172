#!/bin/bash
# call 31: batched streaming forward: tests, probe, config 5 A/B (batch route vs per-frame loop vs per-frame stand-in)
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_video_depth_anything_net.py tests/test_video_depth_anything.py tests/test_depth_anything.py -m gpu -x -q -s 2>&1 | tail -8 > gpurun_out/r05aa_vda_tests.log
echo "rc=$?" >> gpurun_out/r05aa_vda_tests.log
cat gpurun_out/r05aa_vda_tests.log
timeout 300 python tools/vda_probe.py > gpurun_out/r05aa_vda_probe.txt 2>&1
head -6 gpurun_out/r05aa_vda_probe.txt
for leg in batch loop perframe; do
  unset NUNIF_CONFIG5_PERFRAME NUNIF_VDA_BATCH
  if [ $leg = perframe ]; then export NUNIF_CONFIG5_PERFRAME=1; fi
  if [ $leg = loop ]; then export NUNIF_VDA_BATCH=0; fi
  timeout 600 python - > gpurun_out/r05aa_c5_$leg.json 2> gpurun_out/r05aa_c5_$leg.err <<'PY'
import json, torch, bench
rec = bench.config5_record(torch.device("cuda:0"))
print(json.dumps(rec))
PY
  python - <<PY
import json
r = json.load(open("gpurun_out/r05aa_c5_$leg.json"))
print("$leg", r["ms_per_frame"], r["fps"], r.get("depth_net"))
PY
done
