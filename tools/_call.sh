#!/bin/bash
# call 47: the final HEAD: full GPU suite
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05fin4_gpu_suite.log
echo "suite rc=$?" >> gpurun_out/r05fin4_gpu_suite.log
tail -3 gpurun_out/r05fin4_gpu_suite.log
