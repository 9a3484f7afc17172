set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stitch.py tests/test_gpu_waifu2x_api.py -m gpu -x -q > gpurun_out/r05m_tests.log 2>&1; tail -2 gpurun_out/r05m_tests.log
BENCH="python bench.py --no-cpu-baseline --no-host-frames --no-iw3 --no-4k --no-cunet --no-config5 --steps 20 --warmup 5"
for i in 1 2; do
  NUNIF_STITCH_FAST=0 timeout 600 $BENCH > gpurun_out/r05m_base_$i.json 2> gpurun_out/r05m_base_$i.err
  timeout 600 $BENCH > gpurun_out/r05m_new_$i.json 2> gpurun_out/r05m_new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05m_*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, r['value'], [k for k in r['kernel_classes'] if 'stitch' in k['kernel']])
    except Exception as e: print(f, 'ERR', e)
PY
