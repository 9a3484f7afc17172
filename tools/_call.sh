set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_light_inpaint.py tests/test_forward_inpaint.py tests/test_mlbw.py tests/test_abi.py -m gpu -x -q > gpurun_out/r05o_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05o_tests.log
tail -4 gpurun_out/r05o_tests.log
for i in 1 2; do
  NUNIF_INPAINT_MIRROR=0 timeout 400 python tools/config5_probe.py > gpurun_out/r05o_c5_base_$i.json 2> gpurun_out/r05o_c5_base_$i.err
  timeout 400 python tools/config5_probe.py > gpurun_out/r05o_c5_new_$i.json 2> gpurun_out/r05o_c5_new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05o_c5_*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, r['ms_per_frame'], r['fps'])
    except Exception as e: print(f, 'ERR', e)
PY
