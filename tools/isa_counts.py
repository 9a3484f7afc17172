#!/usr/bin/env python3
"""Static instruction mix of the hot kernels: compiles the named csrc files to gfx950 ISA with the build's own flags and counts, per
kernel, VALU / MFMA / LDS / vector-memory / scalar instructions, registers and the s_waitcnt immediates hipcc chose.
    python tools/isa_counts.py [file.hip ...]  > profiles/rNN_isa_counts.txt
Counts are of the STATIC code (loops counted once) — they line up with SQ_INSTS_* only for straight-line kernels."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nunif_amd import build  # noqa: E402

files = sys.argv[1:] or ["swin_qkv_attn_r.hip", "swin_block_tail.hip", "swin_block_tail_ws.hip", "conv3_dma.hip", "conv3_lds.hip",
                         "depth_mlp.hip", "iw3_warp.hip"]
for f in files:
    src = os.path.join(build.CSRC, f)
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        flags = [x for x in build.FLAGS if x != "-fPIC"] + build.EXTRA_FLAGS.get(f, [])
        subprocess.run([build.hipcc()] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", tmp.name, src],
                       check=True, capture_output=True)
        text = open(tmp.name).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    print(f"== {f}")
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        c = collections.Counter()
        waits = collections.Counter()
        for line in body.splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
                continue
            op = t[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("global_load_lds", "buffer_load")) and "lds" in line: c["lds_dma"] += 1
            elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): c["vmem"] += 1
            elif op == "s_waitcnt": waits[" ".join(t[1:])] += 1
            elif op == "s_barrier": c["barrier"] += 1
            elif op == "s_nop": c["s_nop"] += 1
            elif op.startswith("s_"): c["salu"] += 1
        try:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            dem = name
        sg, vg, sp = meta.get(name, (0, 0, 0))
        print(f"  {dem[:110]}")
        print(f"    vgpr {vg} (spilled {sp}) sgpr {sg} | " + " ".join(f"{k} {v}" for k, v in sorted(c.items())))
        top = ", ".join(f"{k} x{v}" for k, v in waits.most_common(6))
        print(f"    s_waitcnt: {top}")
