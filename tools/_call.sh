cd /root/repo
AB_TESTS="tests/test_gpu_swin.py tests/test_gpu_waifu2x_api.py tests/test_gpu_swin_v2.py tests/test_hot_regime.py" bash tools/ab_multi.sh r06o 2 cur t175 t175p3 2>&1 | tail -20
REPO=$PWD; OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --batch-size 45 --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --streams 1 --steps 2 --warmup 1"
for v in t175; do
  d=/tmp/pf_$v; rm -rf $d
  NUNIF_HIP_LIB=$REPO/nunif_amd/libnunif_hip_$v.so rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -o pmc -- $BENCH > /dev/null 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  echo "== $v"; python $REPO/tools/aggregate_pmc.py "$f" FETCH_SIZE | grep "qkv_attn_r_kernel<96" | cut -c1-150
done | tee -a $OUT/r06n_attn_fetch_by_variant.txt
