#!/bin/bash
# end-of-round evidence in ONE gpurun call: rocprofv3 kernel stats + TCC counters of the bench workload, of the iw3 ops and of
# the iw3 scheduler, then the default bench line, the smoke entry point.   bash tools/round_final.sh r02c
TAG=${1:-rXX}
REPO=$(pwd)
bash tools/profile_round.sh $TAG > /dev/null 2>&1
bash tools/profile_iw3_ops.sh $TAG > /dev/null 2>&1
bash tools/aten_census.sh > gpurun_out/${TAG}_aten_census.txt 2>&1; cp gpurun_out/iw3_sched_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_iw3_sched.csv
cd $REPO
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1
tail -2 gpurun_out/${TAG}_smoke.log
ls gpurun_out | grep "^${TAG}"
