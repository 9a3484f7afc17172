set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cunet.py tests/test_convstack.py tests/test_hot_regime.py tests/test_gpu_waifu2x_api.py -m gpu -x -q -k "not swin" > gpurun_out/r05h_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05h_tests.log
tail -4 gpurun_out/r05h_tests.log
for i in 1 2; do
  NUNIF_CUNET_SLICED=0 CUNET_BATCH=66 CUNET_PROF=1 CUNET_ITERS=30 timeout 300 python tools/cunet_probe.py > gpurun_out/r05h_cunet_base_$i.txt 2>&1
  CUNET_BATCH=66 CUNET_PROF=1 CUNET_ITERS=30 timeout 300 python tools/cunet_probe.py > gpurun_out/r05h_cunet_new_$i.txt 2>&1
done
grep -H "MPix" gpurun_out/r05h_cunet_*.txt
head -14 gpurun_out/r05h_cunet_new_2.txt
