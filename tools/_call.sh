set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_depth_anything.py tests/test_depth_anything_vs_hf.py tests/test_cunet.py tests/test_gpu_iw3_pipeline.py tests/test_hot_regime.py -m gpu -x -q -k "not swin" > gpurun_out/r05f_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05f_tests.log
tail -4 gpurun_out/r05f_tests.log
for i in 1 2 3; do
  NUNIF_DA_RCU1_BRANCH=0 timeout 300 python tools/da_probe.py vits > gpurun_out/r05f_da_base_$i.txt 2>&1
  timeout 300 python tools/da_probe.py vits > gpurun_out/r05f_da_new_$i.txt 2>&1
done
grep -H "fps" gpurun_out/r05f_da_*.txt
NUNIF_DA_RCU1_BRANCH=0 timeout 300 python tools/da_probe.py vitb vitl > gpurun_out/r05f_da_base_bl.txt 2>&1
timeout 300 python tools/da_probe.py vitb vitl > gpurun_out/r05f_da_new_bl.txt 2>&1
grep -H "fps" gpurun_out/r05f_da_*_bl.txt
CUNET_BATCH=66 CUNET_PROF=1 CUNET_ITERS=30 timeout 300 python tools/cunet_probe.py > gpurun_out/r05f_cunet.txt 2>&1
grep -H "MPix\|se_block" gpurun_out/r05f_cunet.txt
