# round 6 final evidence: GPU suite, round_final (profiles + default bench line + smoke), 4K PMC set, hot-regime statistics
cd /root/repo
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r06fin_gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $OUT/r06fin_gpu_suite.log)"
bash tools/profile_4k.sh r06k > /dev/null 2>&1
cp $OUT/r06k_pmc_FETCH_SIZE.txt $OUT/r06k_pmc_WRITE_SIZE.txt profiles/ 2>/dev/null
bash tools/round_final.sh r06fin 2>&1 | tail -4
timeout 900 python tools/hot_regime_stats.py 8 > $OUT/r06fin_hot.json 2> $OUT/r06fin_hot.err; tail -4 $OUT/r06fin_hot.err
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r06fin_bench_line.json') if l.startswith('{')][-1])
print('value',r['value'],'single',r['single_stream'],'psnr',r.get('psnr_vs_oracle_db'),r.get('psnr_whole_frame_db'),'mfma',r['model_mfma_frac'])
print('roofline',{k:r['roofline'][k] for k in ('kernel','frac','avg_launch_us','traffic')})
for k in r['kernel_classes'][:6]: print(k)
print('4k',{k:r['scale4x_4k'].get(k) for k in ('ms_per_frame','value','model_mfma_frac')}, r['scale4x_4k'].get('roofline',{}).get('traffic'))
print('cunet',r['cunet']['frame_1080p'],'iw3',{k:v for k,v in r['iw3'].items() if k in ('forward_fill','row_flow_v3','depth_infer_fps')})
print('config5',r['config5'].get('ms_per_frame'),'cpu',r['cpu_baseline'].get('value'),'ok',r['ok'])
PY
