"""Summarise gpurun_out/<tag>_ab_{base,new}_<i>.json (tools/ab_lib.sh): value, single-stream value and the per-kernel table."""
import glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
for f in sorted(glob.glob(os.path.join(out, tag + "*_ab_*_*.json"))):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e, open(f[:-4] + "err").read()[-400:] if os.path.exists(f[:-4] + "err") else "")
        continue
    print(os.path.basename(f), "value", r["value"], "single", (r.get("single_stream") or {}).get("value"), "psnr", r.get("psnr_vs_oracle_db"))
    for k in r["kernel_classes"][:9]:
        print("    %-38s %8.1f us x%d  share %.3f tflops %.0f" % (k["kernel"], k["avg_us"], k["launches_per_frame"], k["share"], k["tflops"]))
