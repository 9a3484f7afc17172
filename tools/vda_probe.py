#!/usr/bin/env python3
"""The Video-Depth-Anything ViT-S streaming network, one frame per call at the reference's default 392-pixel lower bound (392 x 700 for
16:9): ms per frame with the 32-frame window full, beside the per-frame Depth-Anything ViT-S at B = 1 (the same encoder and DPT head
without the temporal modules), and the kernel classes of the temporal part."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import _hip  # noqa: E402
from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2  # noqa: E402
from nunif_amd.iw3.video_depth_anything_net import HipVideoDepthAnythingStreaming  # noqa: E402
from nunif_amd.synthetic import depth_anything_v2_state_dict, video_depth_anything_state_dict  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
h, w = 392, 700
frames = [torch.randn(3, h, w, device=dev) for _ in range(4)]


def timed(fn, n=60):
    for i in range(40):                      # fills the window (and warms the caches / allocations)
        fn(frames[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(frames[i % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


vda = HipVideoDepthAnythingStreaming(video_depth_anything_state_dict(601), dev)
da = HipDepthAnythingV2(depth_anything_v2_state_dict(601), dev)
t_da = timed(lambda f: da(f.unsqueeze(0)))
t_vda = timed(vda.infer_video_depth_one)
print(f"per-frame ViT-S (B = 1, {h} x {w}): {t_da:.3f} ms/frame;  VDA streaming ViT-S: {t_vda:.3f} ms/frame "
      f"(temporal part {t_vda - t_da:+.3f} ms)")
for nb in (3, 4, 8):
    xb = torch.stack([frames[i % 4] for i in range(nb)])
    t_b = timed(lambda f: vda.infer_video_depth_batch(xb), n=30) / nb
    t_d = timed(lambda f: da(xb), n=30) / nb
    print(f"batches of {nb} consecutive frames: VDA streaming {t_b:.3f} ms/frame, per-frame ViT-S {t_d:.3f} ms/frame")
_hip.profile_read(reset=True)
_hip.profile_enable(True)
for i in range(8):
    vda.infer_video_depth_one(frames[i % 4])
torch.cuda.synchronize()
recs = sorted(_hip.profile_read(reset=True), key=lambda r: -r["total_ms"])
_hip.profile_enable(False)
tot = sum(r["total_ms"] for r in recs) / 8
print(f"kernel classes, ms per frame (sum {tot:.3f}):")
for r in recs[:int(os.environ.get("VDA_TOP", "16"))]:
    sec = r["total_ms"] * 1e-3
    print(f"    {r['name'][:34]:34s} {r['total_ms'] / 8:8.4f} ms/frame  {r['launches'] // 8:4d} launches  "
          f"{r['bytes'] / sec / 1e9 if sec else 0:8.1f} GB/s")
