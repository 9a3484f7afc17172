#!/usr/bin/env python3
"""Depth backbones: BaseDepthModel.infer on a 4 x 1080p batch (the reference's _bench protocol), fps per encoder, and the
per-kernel-class table of the ViT-S run."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import _hip  # noqa: E402
from nunif_amd.iw3.base_depth_model import CallableDepthModel  # noqa: E402
from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2  # noqa: E402
from nunif_amd.synthetic import depth_anything_v2_state_dict  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
x4 = torch.rand(4, 3, 1080, 1920, device=dev)
for enc in (sys.argv[1:] or ["vits", "vitb", "vitl"]):
    dm = CallableDepthModel(HipDepthAnythingV2(depth_anything_v2_state_dict(601, grid=37, encoder=enc), dev))
    dm.load(gpu=0)
    for _ in range(2):
        dm.infer(x4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dm.infer(x4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{enc}: {4 / dt:.1f} fps  ({dt * 1e3:.2f} ms per batch of 4)")
    if enc == os.environ.get("DA_PROF", "vits"):
        _hip.profile_read(reset=True)
        _hip.profile_enable(True)
        for _ in range(3):
            dm.infer(x4)
        torch.cuda.synchronize()
        recs = sorted(_hip.profile_read(reset=True), key=lambda r: -r["total_ms"])
        _hip.profile_enable(False)
        for r in recs[:int(os.environ.get("DA_TOP", "10"))]:
            sec = r["total_ms"] * 1e-3
            print(f"    {r['name'][:34]:34s} {r['total_ms'] / 3:8.3f} ms/batch  {r['launches'] // 3:4d} launches  "
                  f"{r['flops'] / sec / 1e12 if sec else 0:7.1f} TF/s")
    del dm
