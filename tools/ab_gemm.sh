#!/bin/bash
# A/B of the PatchUp / PatchDown GEMM variants: per-kernel HIP-event averages of the single-stream leg
run() {
  label=$1; shift
  out=$(env "$@" timeout 120 python bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-host-frames 2>&1 | tail -1)
  echo "$label $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], [(k['kernel'],k['avg_us'],k['launches_per_frame']) for k in d['kernel_classes'] if 'gemm' in k['kernel']])" 2>&1 | tail -1)"
}
run base X=1
run mf2res NUNIF_GEMM6_MF=2
run mf2ring NUNIF_GEMM6_MF=2 NUNIF_GEMM_RING=1
run d2split NUNIF_DOWN2_SPLIT=1
run both NUNIF_GEMM6_MF=2 NUNIF_DOWN2_SPLIT=1
NUNIF_GEMM6_MF=2 NUNIF_DOWN2_SPLIT=1 timeout 200 python -m pytest tests/test_gpu_swin.py -m gpu -x -q -k "golden or 112" 2>&1 | tail -2
