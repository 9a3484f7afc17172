set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05t_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05t_gpu_suite.log
tail -3 gpurun_out/r05t_gpu_suite.log
python bench.py > gpurun_out/r05t_bench_line.json 2> gpurun_out/r05t_bench.err
echo "bench rc=$?"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05t_smoke.log 2>&1
tail -2 gpurun_out/r05t_smoke.log
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r05t_bench_line.json') if l.startswith('{')][-1])
print({k:r.get(k) for k in ('value','ms_per_step','single_stream','psnr_vs_oracle_db','psnr_whole_frame_db','model_mfma_frac','ok','errors')})
print(r['psnr_whole_frame'])
print('cunet', r['cunet']['frame_1080p'], r['cunet']['frame_1080p_batch16'], r['cunet']['model_mfma_frac'])
print('iw3', r['iw3']['forward_fill'], r['iw3']['row_flow_v3'], r['iw3']['depth_infer_fps'], r['iw3']['cpu_baseline'])
print('4k', r['scale4x_4k']['ms_per_frame'], 'config5', r['config5']['ms_per_frame'])
for k in r['kernel_classes']: print(k['kernel'], k['avg_us'])
PY
