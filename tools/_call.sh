#!/bin/bash
# call 32: end-of-round evidence at the final HEAD: full GPU suite, round_final (kernel stats, PMC, bench line, smoke), VDA PMC
cd /root/repo
mkdir -p gpurun_out
REPO=$(pwd); OUT=$REPO/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05fin_gpu_suite.log
echo "suite rc=$?" >> gpurun_out/r05fin_gpu_suite.log
cat gpurun_out/r05fin_gpu_suite.log
timeout 1500 bash tools/round_final.sh r05fin
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pv_$ctr; rm -rf $d
    timeout 180 rocprofv3 --pmc $ctr --output-format csv -d $d -o pmc -- python $REPO/tools/vda_pmc_probe.py > "$OUT/r05v_pmc_$ctr.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $REPO/tools/aggregate_pmc.py "$f" $ctr > "$OUT/r05v_pmc_${ctr}.txt"
done
cd $REPO
grep -h "vda_tattn" gpurun_out/r05v_pmc_*.txt | head
head -c 600 gpurun_out/r05fin_bench_line.json
