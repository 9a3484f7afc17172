set -u
mkdir -p gpurun_out
for i in 1 2; do
NUNIF_SHARD_LAG=0 LEG_FRAMES=96 timeout 600 python tools/bench_legs_smoke.py > gpurun_out/r05s_legs_lag0_$i.json 2> gpurun_out/r05s_legs_lag0_$i.err
NUNIF_SHARD_LAG=1 LEG_FRAMES=96 timeout 600 python tools/bench_legs_smoke.py > gpurun_out/r05s_legs_lag1_$i.json 2> gpurun_out/r05s_legs_lag1_$i.err
NUNIF_SHARD_LAG=2 LEG_FRAMES=96 timeout 600 python tools/bench_legs_smoke.py > gpurun_out/r05s_legs_lag2_$i.json 2> gpurun_out/r05s_legs_lag2_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05s_legs_*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, r['iw3']['fps'], r['iw3']['frames_delivered'], r['cunet']['value'])
    except Exception as e: print(f,'ERR',e)
PY
