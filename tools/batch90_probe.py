#!/usr/bin/env python3
"""Is a 90-tile launch (two 1080p frames' worth of tiles in one minibatch) more efficient per tile than two 45-tile launches?
Proxy: a 2160 x 1920 frame (10 x 9 tiles) at batch 90 vs a 1080 x 1920 frame (5 x 9 tiles) at batch 45, one stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import _hip  # noqa: E402
from nunif_amd.nunif.utils.render import tiled_render  # noqa: E402
from nunif_amd.synthetic import swin_unet_state_dict  # noqa: E402
from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = SwinUNet2x().eval()
m.load_state_dict(swin_unet_state_dict(102, 2))
m = m.to(dev)
for (H, W, b) in ((1080, 1920, 45), (2160, 1920, 90), (2160, 1920, 45), (3240, 1920, 135)):
    x = torch.rand(3, H, W, device=dev)
    for _ in range(3):
        tiled_render(x, m, tile_size=256, batch_size=b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        tiled_render(x, m, tile_size=256, batch_size=b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{H}x{W} batch {b}: {dt * 1e3:.3f} ms  = {H * W / dt / 1e6:.1f} MPix/s")
