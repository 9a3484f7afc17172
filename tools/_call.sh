set -u
mkdir -p gpurun_out
for i in 1 2; do
  NUNIF_PATCHUP=0 NUNIF_PATCHDOWN=0 timeout 300 python tools/bench_4k4x.py > gpurun_out/r05n_4k_base_$i.txt 2>&1
  NUNIF_PATCHDOWN=0 timeout 300 python tools/bench_4k4x.py > gpurun_out/r05n_4k_up_$i.txt 2>&1
  timeout 300 python tools/bench_4k4x.py > gpurun_out/r05n_4k_new_$i.txt 2>&1
done
grep -H "batch" gpurun_out/r05n_4k_*.txt
