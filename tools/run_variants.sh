python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 8 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('psnr_vs_oracle_db'))
for k in j['kernel_classes']: print('  ', k['kernel'], k['avg_us'], k['launches_per_frame'], k['tflops'], k['gbs'])"
