cd /root/repo
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r06f_gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $OUT/r06f_gpu_suite.log)"
for i in 1 2; do
  NUNIF_HIP_LIB=$PWD/nunif_amd/libnunif_hip_head.so timeout 200 python tools/scale4x_probe.py 2>/dev/null | tail -1 | sed 's/^/head: /'
  timeout 200 python tools/scale4x_probe.py 2>/dev/null | tail -1 | sed 's/^/new:  /'
done | tee $OUT/r06f_4k_toimage_ab.txt
HOT_TAPS_INPUTS=4 timeout 600 python tools/hot_taps.py 2x_chaos 2x > $OUT/r06_hot_taps.txt 2> $OUT/r06_hot_taps.err; tail -3 $OUT/r06_hot_taps.err; grep -n "first stage\|<--" $OUT/r06_hot_taps.txt
bash tools/profile_sq.sh r06b > /dev/null 2>&1; grep -A2 "SQ_INSTS_VALU$\|SQ_INSTS_MFMA$\|SQ_LDS_BANK_CONFLICT$" $OUT/r06b_sq.txt | grep "qkv_attn_r_kernel<96" | cut -c1-160
