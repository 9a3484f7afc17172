#!/usr/bin/env python3
"""waifu2x cunet / upcunet whole-frame render of a 1080p frame (tile 256): ms per frame."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd.nunif.models import create_model  # noqa: E402
from nunif_amd.nunif.utils.render import tiled_render  # noqa: E402
import nunif_amd.waifu2x.utils  # noqa: E402,F401
from nunif_amd.synthetic import cunet_state_dict  # noqa: E402

torch.set_grad_enabled(False)
BATCH = int(os.environ.get("CUNET_BATCH", "16"))
TAGS = os.environ.get("NUNIF_PROF_TAGS")
ONLY = os.environ.get("CUNET_ONLY")
ITERS = int(os.environ.get("CUNET_ITERS", "20"))
x = torch.rand(3, 1080, 1920, device="cuda:0")
for name, up in (("waifu2x.cunet", False), ("waifu2x.upcunet", True)):
    if ONLY and not name.endswith("." + ONLY):
        continue
    m = create_model(name).eval()
    m.load_state_dict(cunet_state_dict(7, up=up))
    m = m.to("cuda:0")
    for _ in range(3):
        tiled_render(x, m, tile_size=256, batch_size=BATCH)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ITERS):
        tiled_render(x, m, tile_size=256, batch_size=BATCH)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / ITERS
    print(f"{name} batch {BATCH}: {dt * 1e3:.3f} ms per 1080p frame = {1080 * 1920 / dt / 1e6:.1f} MPix/s")
    if os.environ.get("CUNET_PROF"):
        from nunif_amd import _hip
        _hip.profile_read(reset=True)
        _hip.profile_enable(True)
        for _ in range(3):
            tiled_render(x, m, tile_size=256, batch_size=BATCH)
        torch.cuda.synchronize()
        recs = sorted(_hip.profile_read(reset=True), key=lambda r: -r["total_ms"])
        _hip.profile_enable(False)
        for r in recs[:(40 if TAGS else 9)]:
            sec = r["total_ms"] * 1e-3
            print(f"    {r['name'][:34]:34s} {r['total_ms'] / 3:8.3f} ms/frame  {r['launches'] // 3:4d} launches  "
                  f"{r['flops'] / sec / 1e12 if sec else 0:7.1f} TF/s  {r['bytes'] / sec / 1e9 if sec else 0:7.0f} GB/s")
