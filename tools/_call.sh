# the very last HEAD of round 6: GPU suite + the default bench line + smoke
cd /root/repo
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r06fin2_gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $OUT/r06fin2_gpu_suite.log)"
python bench.py > $OUT/r06fin2_bench_line.json 2> $OUT/r06fin2_bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r06fin2_smoke.log 2>&1; tail -1 $OUT/r06fin2_smoke.log
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r06fin2_bench_line.json') if l.startswith('{')][-1])
print('value',r['value'],'single',r['single_stream']['value'],'psnr',r.get('psnr_whole_frame_db'),'roofline',r['roofline']['frac'],r['roofline']['avg_launch_us'])
print([ (k['kernel'],k['avg_us']) for k in r['kernel_classes'][:4]])
print('4k',r['scale4x_4k']['ms_per_frame'],'cunet',r['cunet']['frame_1080p']['value'],'ff',r['iw3']['forward_fill']['fps'],'vits',r['iw3']['depth_infer_fps']['vits'],'c5',r['config5']['ms_per_frame'],'ok',r['ok'])
PY
