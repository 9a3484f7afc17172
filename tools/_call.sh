set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cunet.py tests/test_light_inpaint.py tests/test_hot_regime.py -m gpu -x -q -k "not swin" > gpurun_out/r05j_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05j_tests.log
tail -4 gpurun_out/r05j_tests.log
for i in 1 2; do
  NUNIF_CUNET_UP=0 CUNET_BATCH=66 CUNET_PROF=1 CUNET_ITERS=30 timeout 300 python tools/cunet_probe.py > gpurun_out/r05j_cunet_base_$i.txt 2>&1
  CUNET_BATCH=66 CUNET_PROF=1 CUNET_ITERS=30 timeout 300 python tools/cunet_probe.py > gpurun_out/r05j_cunet_new_$i.txt 2>&1
done
grep -H "MPix" gpurun_out/r05j_cunet_*.txt
head -12 gpurun_out/r05j_cunet_new_2.txt
NUNIF_PATCHUP=0 timeout 400 python tools/config5_probe.py > gpurun_out/r05j_c5_base.json 2> gpurun_out/r05j_c5_base.err
timeout 400 python tools/config5_probe.py > gpurun_out/r05j_c5_new.json 2> gpurun_out/r05j_c5_new.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05j_c5_*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, r['ms_per_frame'], r['fps'])
    except Exception as e: print(f, 'ERR', e)
PY
