#!/bin/bash
# call 34: the process-level A/B switches (subprocess probe) + the cunet head switches
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ab_switches.py tests/test_cunet.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05ac_switch_tests.log
echo "rc=$?" >> gpurun_out/r05ac_switch_tests.log
cat gpurun_out/r05ac_switch_tests.log
