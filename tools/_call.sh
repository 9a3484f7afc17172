set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05k_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05k_gpu_suite.log
tail -3 gpurun_out/r05k_gpu_suite.log
for i in 1 2; do
  NUNIF_CUNET_UP=0 CUNET_BATCH=66 CUNET_ITERS=30 CUNET_ONLY=cunet timeout 300 python tools/cunet_probe.py > gpurun_out/r05k_cunet_base_$i.txt 2>&1
  CUNET_BATCH=66 CUNET_ITERS=30 CUNET_ONLY=cunet CUNET_PROF=1 timeout 300 python tools/cunet_probe.py > gpurun_out/r05k_cunet_new_$i.txt 2>&1
done
grep -H "MPix\|cunet_up" gpurun_out/r05k_cunet_*.txt
