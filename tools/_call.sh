set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_swin.py tests/test_cunet.py tests/test_gpu_waifu2x_api.py -m gpu -x -q > gpurun_out/r05u_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05u_tests.log; tail -3 gpurun_out/r05u_tests.log
BENCH="python bench.py --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --steps 40 --warmup 5"
for i in 1 2; do
  NUNIF_STEM_ROWS=2 timeout 600 $BENCH > gpurun_out/r05u_ab_base_$i.json 2> gpurun_out/r05u_ab_base_$i.err
  timeout 600 $BENCH > gpurun_out/r05u_ab_new_$i.json 2> gpurun_out/r05u_ab_new_$i.err
  NUNIF_STEM_ROWS_CUNET=2 CUNET_BATCH=66 CUNET_ITERS=30 CUNET_ONLY=cunet CUNET_PROF=1 timeout 300 python tools/cunet_probe.py > gpurun_out/r05u_cunet_base_$i.txt 2>&1
  CUNET_BATCH=66 CUNET_ITERS=30 CUNET_ONLY=cunet CUNET_PROF=1 timeout 300 python tools/cunet_probe.py > gpurun_out/r05u_cunet_new_$i.txt 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05u_ab_*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, r['value'], r['single_stream']['value'], [ (k['kernel'],k['avg_us']) for k in r['kernel_classes'] if 'stem' in k['kernel']])
    except Exception as e: print(f, 'ERR', e)
PY
grep -H "MPix\|stem" gpurun_out/r05u_cunet_*.txt
