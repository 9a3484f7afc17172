#!/bin/bash
# call 25: the Video-Depth-Anything temporal network: parity tests, probe, config 5 with it
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_video_depth_anything_net.py -m gpu -x -q -s 2>&1 | tail -30 > gpurun_out/r05v_vda_tests.log
echo "rc=$?" >> gpurun_out/r05v_vda_tests.log
cat gpurun_out/r05v_vda_tests.log
timeout 300 python tools/vda_probe.py > gpurun_out/r05v_vda_probe.txt 2>&1
cat gpurun_out/r05v_vda_probe.txt | tail -25
