#!/bin/bash
# Same-box A/B of the depth backbones: tests on the product library, then tools/da_probe.py alternating <base variant> / product.
#     bash tools/ab_depth.sh <tag> <base variant name> [encoders...]
set -u
TAG=$1; BASE=$2; shift 2
ENC=${@:-vits}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_depth_anything.py tests/test_depth_anything_vs_hf.py tests/test_video_depth_anything_net.py tests/test_gpu_iw3_pipeline.py -m gpu -x -q > "$OUT/${TAG}_depth_tests.log" 2>&1
echo "depth tests rc=$? $(tail -1 "$OUT/${TAG}_depth_tests.log")" | tee "$OUT/${TAG}_depth_ab.txt"
for i in 1 2; do
    echo "== base ($BASE) run $i" >> "$OUT/${TAG}_depth_ab.txt"
    NUNIF_HIP_LIB=$REPO/nunif_amd/libnunif_hip_$BASE.so DA_TOP=8 timeout 300 python tools/da_probe.py $ENC >> "$OUT/${TAG}_depth_ab.txt" 2>&1
    echo "== new run $i" >> "$OUT/${TAG}_depth_ab.txt"
    DA_TOP=8 timeout 300 python tools/da_probe.py $ENC >> "$OUT/${TAG}_depth_ab.txt" 2>&1
done
cat "$OUT/${TAG}_depth_ab.txt"
