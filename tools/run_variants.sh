python -m pytest tests/test_gpu_swin.py -m gpu -x -q -s 2>&1 | grep -E "PSNR|passed|failed" | tail -4
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['psnr_vs_oracle_db'], j['roofline'])
for k in j['kernel_classes']: print('  ', k['kernel'], k['avg_us'], k['launches_per_frame'], k['tflops'], k['gbs'])"
