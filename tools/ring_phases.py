#!/usr/bin/env python3
"""Per-phase host timing of one frame through the PCIe-inclusive path with the REAL render (serial, one frame at a time):
host write of the pinned input, launch of the zero-copy in-edge + render + out-edge, sync, host read."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
h_in = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
h_out = torch.empty((2 * H, 2 * W, 3), dtype=torch.uint8).pin_memory()
src = np.random.randint(0, 256, (H, W, 3), dtype=np.uint8)
use_side = os.environ.get("SIDE", "1") == "1"
st = torch.cuda.Stream(dev) if use_side else torch.cuda.current_stream(dev)
rows = []
for i in range(24):
    t = [time.perf_counter()]
    np.copyto(h_in.numpy(), src); t.append(time.perf_counter())
    with torch.cuda.stream(st):
        x = _ops.frame_to_tensor(h_in, device=dev); t.append(time.perf_counter())
        y = tiled_render(x, m, tile_size=256, batch_size=45); t.append(time.perf_counter())
        _ops.to_frame(y, 8, out=h_out); t.append(time.perf_counter())
    if os.environ.get("SYNC", "stream") == "stream":
        st.synchronize()
    elif os.environ["SYNC"] == "event":
        ev = torch.cuda.Event(); ev.record(st); ev.synchronize()
    elif os.environ["SYNC"] == "event_blocking":
        ev = torch.cuda.Event(blocking=True); ev.record(st); ev.synchronize()
    elif os.environ["SYNC"] == "query":
        ev = torch.cuda.Event(); ev.record(st)
        while not ev.query():
            pass
    t.append(time.perf_counter())
    _ = int(h_out.numpy()[::64, ::64].sum()); t.append(time.perf_counter())
    rows.append([(t[k + 1] - t[k]) * 1e3 for k in range(6)])
r = np.array(rows[4:])
names = ["host write", "in-edge launch", "render launch", "out-edge launch", "sync", "host read"]
print("side stream" if use_side else "default stream", "sync =", os.environ.get("SYNC", "stream"))
for k, n in enumerate(names):
    print(f"  {n:16s} median {np.median(r[:, k]):7.2f} ms  max {r[:, k].max():7.2f}")
print(f"  total            median {np.median(r.sum(1)):7.2f} ms  max {r.sum(1).max():7.2f}")
