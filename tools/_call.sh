#!/bin/bash
# call 28: temporal attention with the whole window in flight: parity tests + probe
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_video_depth_anything_net.py -m gpu -x -q -s 2>&1 | tail -8 > gpurun_out/r05y_vda_tests.log
echo "rc=$?" >> gpurun_out/r05y_vda_tests.log
cat gpurun_out/r05y_vda_tests.log
timeout 300 python tools/vda_probe.py > gpurun_out/r05y_vda_probe.txt 2>&1
tail -20 gpurun_out/r05y_vda_probe.txt
