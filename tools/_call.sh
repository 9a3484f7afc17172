#!/bin/bash
# call 37: the cunet switch test alone, then the full GPU suite
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cunet.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05fin2_cunet_test.log
cat gpurun_out/r05fin2_cunet_test.log | tail -4
if grep -q "failed" gpurun_out/r05fin2_cunet_test.log; then exit 0; fi
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05fin2_gpu_suite.log
echo "suite rc=$?" >> gpurun_out/r05fin2_gpu_suite.log
tail -4 gpurun_out/r05fin2_gpu_suite.log
