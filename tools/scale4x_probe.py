#!/usr/bin/env python3
"""BASELINE configs[2] on one GPU: waifu2x swin_unet 4x on a 4K frame (170 tiles of 256, minibatches of 34): ms per frame and, with
SCALE4X_PROF=1, the per-kernel classes (HIP events).  The command tools/profile_4k.sh hands to rocprofv3."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
if os.environ.get("SCALE4X_PROF"):
    print(json.dumps(bench.scale4x_record(dev)))
else:
    from nunif_amd.nunif.utils.render import tiled_render
    from nunif_amd.synthetic import swin_unet_state_dict
    from nunif_amd.waifu2x.models.swin_unet import SwinUNet4x
    m = SwinUNet4x().eval()
    m.load_state_dict(swin_unet_state_dict(104, 4))
    m = m.to(dev)
    x = bench.synth_frame(77, 2160, 3840).to(dev)
    iters = int(os.environ.get("SCALE4X_ITERS", "3"))
    tiled_render(x, m, tile_size=256, batch_size=34)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tiled_render(x, m, tile_size=256, batch_size=34)
    torch.cuda.synchronize()
    print(f"swin_unet 4x, 4K frame: {(time.perf_counter() - t0) / iters * 1e3:.2f} ms per frame")
