#!/usr/bin/env python3
"""Build container only: the REAL reference (``/root/reference`` through ``oracle/refstub.py``: ``nunif.utils.render.tiled_render``
+ the reference's ``SwinUNet2x`` over the pinned swin block) timed next to the oracle port (``oracle.seam_blending.tiled_render``
+ ``oracle.swin_unet.model_forward``) on the same crop, same weights, same thread count — the ratio that turns bench.py's
``cpu_baseline`` (kind "port", the only thing that can run on the GPU box) into an estimate of the reference's own CPU path
(BASELINE.md §4).  Prints one JSON line; the numbers go to DESIGN.md §5."""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refstub  # noqa: E402

refstub.install()
torch.set_grad_enabled(False)


def med(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), out


def main():
    from nunif.utils.render import tiled_render as ref_render
    from waifu2x.models.swin_unet import SwinUNet2x
    from oracle import seam_blending as OS
    from oracle import swin_unet as O
    from nunif_amd.synthetic import swin_unet_state_dict
    threads = int(os.environ.get("THREADS", os.cpu_count() or 8))
    torch.set_num_threads(threads)
    sd = swin_unet_state_dict(102, 2)
    m = SwinUNet2x().eval()
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(3, 476, 476, generator=g)
    out = {"threads": threads, "crop": [476, 476], "tiles": 4}
    for bs in (1, 4):
        t_ref, y_ref = med(lambda: ref_render(x, m, tile_size=256, batch_size=bs))
        t_port, y_port = med(lambda: OS.tiled_render(x, lambda mb: O.model_forward(sd, mb), 2, 16, 8, 256, bs))
        out[f"batch{bs}"] = {"reference_s": round(t_ref, 3), "port_s": round(t_port, 3), "port_over_reference": round(t_port / t_ref, 3),
                             "max_abs_diff": float((y_ref - y_port).abs().max())}
    if "--whole-frame" in sys.argv:
        # BASELINE.md section 4's protocol on the reference itself: ONE WHOLE 1080p frame (45 tiles), warm, timed once each
        low = torch.rand(1, 3, 1080 // 16 + 1, 1920 // 16 + 1, generator=g)
        frame = torch.nn.functional.interpolate(low, size=(1080, 1920), mode="bilinear", align_corners=False)[0].clamp(0, 1)
        t0 = time.perf_counter()
        y_ref = ref_render(frame, m, tile_size=256, batch_size=1)
        t_ref = time.perf_counter() - t0
        t0 = time.perf_counter()
        y_port = OS.tiled_render(frame, lambda mb: O.model_forward(sd, mb), 2, 16, 8, 256, 1)
        t_port = time.perf_counter() - t0
        out["frame_1080p"] = {"tiles": 45, "reference_s": round(t_ref, 2), "port_s": round(t_port, 2),
                              "reference_mpix_per_s": round(1080 * 1920 / 1e6 / t_ref, 4), "port_over_reference": round(t_port / t_ref, 3),
                              "max_abs_diff": float((y_ref - y_port).abs().max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
