#!/bin/bash
# call 46: forward warp with one pair per thread for rows up to 2 048 pixels: iw3 tests + A/B
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_iw3.py tests/test_gpu_iw3_pipeline.py tests/test_gpu_iw3_scheduler.py tests/test_forward_inpaint.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05fin4_iw3_tests.log
cat gpurun_out/r05fin4_iw3_tests.log
( for f in 0 1 0 1; do
  NUNIF_FW_DIET=$f timeout 120 python tools/bench_fw.py 200 2>&1 | grep forward_fill | sed "s/^/diet $f: /"
done ) | tee gpurun_out/r05fin4_fw_diet.txt
