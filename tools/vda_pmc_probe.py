#!/usr/bin/env python3
"""Workload for the TCC-counter passes of the Video-Depth-Anything streaming network (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
331 frames at 392 x 700, one per call — 300 of them with the 32-frame window full, so that the per-launch average of
`vda_tattn2_kernel` is that of a full window to within ~5 %."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd.iw3.video_depth_anything_net import HipVideoDepthAnythingStreaming  # noqa: E402
from nunif_amd.synthetic import video_depth_anything_state_dict  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
frames = [torch.randn(3, 392, 700, device=dev) for _ in range(4)]
vda = HipVideoDepthAnythingStreaming(video_depth_anything_state_dict(601), dev)
for i in range(331):
    vda.infer_video_depth_one(frames[i % 4])
torch.cuda.synchronize()
print("done")
