set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_swin.py tests/test_gpu_waifu2x_api.py tests/test_gpu_swin_v2.py -m gpu -x -q > gpurun_out/r05c_swin_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05c_swin_tests.log
tail -4 gpurun_out/r05c_swin_tests.log
BENCH="python bench.py --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --steps 60 --warmup 5"
for i in 1 2; do
  NUNIF_PATCHDOWN=0 timeout 600 $BENCH > gpurun_out/r05c_ab_base_$i.json 2> gpurun_out/r05c_ab_base_$i.err
  timeout 600 $BENCH > gpurun_out/r05c_ab_new_$i.json 2> gpurun_out/r05c_ab_new_$i.err
  NUNIF_HIP_LIB=$PWD/nunif_amd/libnunif_hip_pukd6.so timeout 600 $BENCH > gpurun_out/r05c_ab_kd6_$i.json 2> gpurun_out/r05c_ab_kd6_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05c_ab_*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, r['value'], r['single_stream']['value'])
        for k in r['kernel_classes']:
            if 'gemm' in k['kernel'] or 'patch' in k['kernel']: print('   ', k)
    except Exception as e: print(f, 'ERR', e)
PY
