for v in 8 16 8 16; do
rm -f /tmp/done
(NUNIF_ATTN_WAVES=$v python bench.py --no-cpu-baseline --no-host-frames --steps 1200 > gpurun_out/b_w$v.log 2>&1; touch /tmp/done) &
for i in $(seq 1 200); do [ -f /tmp/done ] && break; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" | sed 's/.*(\([0-9]*\)Mhz)/sclk \1/; s/.*Power (W): /W /' | tr "\n" " "; echo; sleep 0.5; done > gpurun_out/smi_w$v.log
wait
python - <<PY
import re,json
rows=[l.split() for l in open("gpurun_out/smi_w$v.log") if l.strip()]
rows=[(int(r[1]),float(r[3])) for r in rows if len(r)>=4 and int(r[1])>1000]
rows=rows[2:-1]
d=json.loads(open("gpurun_out/b_w$v.log").read().strip().splitlines()[-1])
print("waves $v", "n",len(rows),"sclk %.0f"%(sum(r[0] for r in rows)/len(rows)),"W %.0f"%(sum(r[1] for r in rows)/len(rows)), d["value"], d["ms_per_step"])
PY
done
rocm-smi --showpowercap 2>&1 | grep -i cap
