#!/usr/bin/env python3
"""Host-side timeline of FrameRing (depth 3): time spent in each part of submit()/_collect() per frame."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import frame_ring as FR  # noqa: E402
from nunif_amd.iw3 import _ops  # noqa: E402
from nunif_amd.nunif.utils.render import tiled_render  # noqa: E402
from nunif_amd.synthetic import swin_unet_state_dict  # noqa: E402
from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 1080, 1920
m = SwinUNet2x().eval()
m.load_state_dict(swin_unet_state_dict(102, 2))
m = m.to(dev)
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return r
    return w


_ops.frame_to_tensor = timed("frame_to_tensor launch", _ops.frame_to_tensor)
_ops.to_frame = timed("to_frame launch", _ops.to_frame)
render = timed("render launch", lambda x: tiled_render(x, m, tile_size=256, batch_size=45))
depth = int(os.environ.get("DEPTH", "3"))
ring = FR.FrameRing(render, (H, W, 3), (2 * H, 2 * W, 3), device=dev, depth=depth, out_mode="view")
ring._collect = timed("_collect (event sync)", ring._collect)
for s in ring.slots:
    s["h_in"].copy_ = timed("h_in.copy_", s["h_in"].copy_)
host = [np.random.randint(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(4)]
for i in range(4):
    ring.submit(host[i % 4])
ring.drain()
acc.clear()
t0 = time.perf_counter()
n = 30
sub = []
for i in range(n):
    t1 = time.perf_counter()
    ring.submit(host[i % 4])
    sub.append((time.perf_counter() - t1) * 1e3)
ring.drain()
dt = (time.perf_counter() - t0) / n * 1e3
print(f"depth {depth}: {dt:.2f} ms per frame; submit() median {np.median(sub):.2f} max {max(sub):.2f}")
for k, v in acc.items():
    v = np.array(v)
    print(f"  {k:24s} n={len(v):3d} median {np.median(v):7.2f} ms  max {v.max():7.2f}  sum/frame {v.sum() / n:7.2f}")
