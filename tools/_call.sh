#!/bin/bash
# call 48: the switch probe with the forward warp in it
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ab_switches.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05fin5_switch_test.log
cat gpurun_out/r05fin5_switch_test.log | tail -6
