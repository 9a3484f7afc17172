set -u
mkdir -p gpurun_out
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
echo "bench rc=$?"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_smoke.log 2>&1
tail -2 gpurun_out/r05_smoke.log
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r05_bench_line.json') if l.startswith('{')][-1])
print({k:r.get(k) for k in ('value','ms_per_step','single_stream','psnr_vs_oracle_db','psnr_whole_frame_db','model_mfma_frac','ok','errors')})
print(r['roofline'])
for k in r['kernel_classes']: print(k)
print('cunet', {k:v for k,v in r['cunet'].items() if k not in ('kernel_classes','config')})
print('iw3', {k:v for k,v in r['iw3'].items() if k not in ('config',)})
print('4k', r.get('scale4x_4k'))
print('config5', {k:v for k,v in r['config5'].items() if k not in ('kernel_classes','config')})
print('cpu', r['cpu_baseline'])
print('host', r['host_frames'])
PY
