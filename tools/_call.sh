set -u
mkdir -p gpurun_out
TAG=r05
bash tools/profile_round.sh $TAG > /dev/null 2>&1
bash tools/profile_iw3_ops.sh $TAG > /dev/null 2>&1
bash tools/aten_census.sh > gpurun_out/${TAG}_aten_census.txt 2>&1; cp gpurun_out/iw3_sched_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_iw3_sched.csv
bash tools/profile_cunet.sh ${TAG}c > /dev/null 2>&1
bash tools/profile_config5.sh ${TAG}f > /dev/null 2>&1
ls gpurun_out | grep "^${TAG}"
