#!/bin/bash
# call 45: the final HEAD with the forward-warp diet: warp tests + A/B, full GPU suite, default bench line, smoke, iw3-ops counters
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_iw3.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05fin3_fw_tests.log
cat gpurun_out/r05fin3_fw_tests.log
( for f in 0 1 0 1; do
  NUNIF_FW_DIET=$f timeout 120 python tools/bench_fw.py 200 2>&1 | grep forward_fill | sed "s/^/diet $f: /"
done ) | tee gpurun_out/r05fin3_fw_diet.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05fin3_gpu_suite.log
echo "suite rc=$?" >> gpurun_out/r05fin3_gpu_suite.log
tail -3 gpurun_out/r05fin3_gpu_suite.log
timeout 900 python bench.py > gpurun_out/r05fin3_bench_line.json 2> gpurun_out/r05fin3_bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05fin3_smoke.log 2>&1
tail -2 gpurun_out/r05fin3_smoke.log
head -c 200 gpurun_out/r05fin3_bench_line.json; echo
timeout 400 bash tools/profile_iw3_ops.sh r05fin3 > /dev/null 2>&1
grep -h "forward_warp" gpurun_out/r05fin3i_pmc_*.txt
