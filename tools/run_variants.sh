python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for v in "A=1" "NUNIF_GEMM_RING=1"; do
  env $v python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bv.log 2>&1
  python - <<PY
import json
f="gpurun_out/bv.log"
try:
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print("variant $v", j["value"], " ".join(f'{k["kernel"]}={k["avg_us"]}' for k in j["kernel_classes"] if "gemm" in k["kernel"] or "stem" in k["kernel"]))
except Exception as e: print("variant $v ERR", e, open(f).read()[-300:])
PY
done
