#!/usr/bin/env python3
"""Per-op timing of the HBM-bound kernels at the BASELINE sizes (1080p / 4K), with the algorithmic bytes of
SURVEY.md §8d.  Prints one JSON object; used for DESIGN.md §5 and profiles/.  Not the contract benchmark (bench.py)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import _hip  # noqa: E402
from nunif_amd.iw3 import _ops  # noqa: E402
from nunif_amd.iw3.forward_warp import apply_divergence_forward_warp  # noqa: E402
from nunif_amd.iw3.backward_warp import apply_divergence_grid_sample  # noqa: E402
from nunif_amd.iw3.dilation import dilate_edge  # noqa: E402
from nunif_amd.iw3.depth_anything_model import batch_preprocess  # noqa: E402
from nunif_amd.nunif.utils.seam_blending import SeamBlending  # noqa: E402
from nunif_amd.synthetic import synth_depth  # noqa: E402

DEV = "cuda:0"
HBM = 8000.0


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    torch.set_grad_enabled(False)
    _hip.lib()
    res = {}

    def rec(name, sec, nbytes, px):
        res[name] = {"us": round(sec * 1e6, 1), "GBps": round(nbytes / sec / 1e9, 1), "frac_hbm": round(nbytes / sec / 1e9 / HBM, 4),
                     "MPix_s": round(px / sec / 1e6, 1)}

    for tag, (H, W) in (("1080p", (1080, 1920)), ("4k", (2160, 3840))):
        B = 2 if tag == "1080p" else 1
        c = torch.rand(B, 3, H, W, device=DEV)
        d = synth_depth(1, B, H, W, "smooth_edges").to(DEV)
        px = B * H * W
        rec(f"forward_warp_fill_{tag}", timeit(lambda: apply_divergence_forward_warp(c, d, 2.0, 0.5, method="forward_fill", width_base=False)), px * 40.0, px)
        rec(f"backward_warp_{tag}", timeit(lambda: apply_divergence_grid_sample(c, d, 2.0, 0.5, "both")), px * 40.0, px)
        l, r = c[0], c[0].flip(-1).contiguous()
        rec(f"stereo_to_frame_u8_{tag}", timeit(lambda: _ops.stereo_to_frame(l, r, "sbs")), H * W * 30.0, H * W)
        dh, dw = (392, 686) if tag == "1080p" else (392, 686)
        ds = synth_depth(2, B, dh, dw, "smooth_edges").to(DEV) * 5
        rec(f"depth_resize_to_frame_{tag}", timeit(lambda: _ops.resize_aa(ds, (H, W), mode="bilinear", align_corners=True)), B * (dh * dw + H * W) * 4.0, px)
        rec(f"batch_preprocess_{tag}", timeit(lambda: batch_preprocess(c)), B * 3 * (H * W + dh * dw) * 4.0, px)
        rec(f"dilate_edge_2_1_{tag}", timeit(lambda: dilate_edge(ds, [2, 1])), B * dh * dw * 16.0 * 2, B * dh * dw)
    # stitch: 1080p 2x (config 2) and 4K 4x (config 3)
    for tag, (H, W, s, off, blend) in (("1080p_2x", (1080, 1920, 2, 16, 8)), ("4k_4x", (2160, 3840, 4, 32, 16))):
        sb = SeamBlending((3, H, W), s, off, 256, blend)
        store = sb._store(torch.device(DEV))
        store.uniform_()
        out_px = sb.y_h * sb.y_w
        rec(f"stitch_{tag}", timeit(lambda: sb.get_output(), iters=10), out_px * 24.5, H * W)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
