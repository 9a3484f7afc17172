set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cunet.py tests/test_gpu_swin.py -m gpu -x -q > gpurun_out/r05i_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05i_tests.log
tail -4 gpurun_out/r05i_tests.log
for i in 1 2; do
  NUNIF_PATCHDOWN=0 CUNET_BATCH=66 CUNET_PROF=1 CUNET_ITERS=30 timeout 300 python tools/cunet_probe.py > gpurun_out/r05i_cunet_base_$i.txt 2>&1
  CUNET_BATCH=66 CUNET_PROF=1 CUNET_ITERS=30 timeout 300 python tools/cunet_probe.py > gpurun_out/r05i_cunet_new_$i.txt 2>&1
done
grep -H "MPix" gpurun_out/r05i_cunet_*.txt
head -12 gpurun_out/r05i_cunet_new_2.txt
