#!/bin/bash
# call 30: cunet head with 16 x 16 tiles vs 16 x 32
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cunet.py tests/test_convstack.py tests/test_gpu_waifu2x_api.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05z2_cunet_tests.log
echo "rc=$?" >> gpurun_out/r05z2_cunet_tests.log
cat gpurun_out/r05z2_cunet_tests.log
for i in 1 2; do
  NUNIF_CUNET_HEAD_TW=32 CUNET_ONLY=cunet CUNET_BATCH=66 CUNET_PROF=1 timeout 300 python tools/cunet_probe.py > gpurun_out/r05z2_cunet_tw32_$i.txt 2>&1
  CUNET_ONLY=cunet CUNET_BATCH=66 CUNET_PROF=1 timeout 300 python tools/cunet_probe.py > gpurun_out/r05z2_cunet_tw16_$i.txt 2>&1
done
grep -H "per 1080p\|head" gpurun_out/r05z2_cunet_tw*.txt
