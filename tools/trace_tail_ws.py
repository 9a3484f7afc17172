#!/usr/bin/env python3
"""Per-trip timeline of the weight-stationary C = 192 swin tail (workgroup 0, all 8 waves), from s_memtime stamps.
Needs an ablation build (NUNIF_BUILD_ABL=1 python -m nunif_amd.build) and NUNIF_TAIL_WS_ABL=256 (+1 / +2 / +3)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import _hip                                           # noqa: E402
from nunif_amd.nunif.utils.render import tiled_render                # noqa: E402
from nunif_amd.synthetic import swin_unet_state_dict                 # noqa: E402
from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x            # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
lib = _hip.lib()
lib.nunif_dbg_ws_trace.restype = ctypes.c_int
lib.nunif_dbg_ws_trace.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8, 64, 8, dtype=torch.int64, device=dev)
assert lib.nunif_dbg_ws_trace(buf.data_ptr()) == 0
m = SwinUNet2x().eval()
m.load_state_dict(swin_unet_state_dict(102, 2))
m = m.to(dev)
x = torch.rand(3, 1080, 1920, device=dev)
for _ in range(3):
    tiled_render(x, m, tile_size=256, batch_size=45)
torch.cuda.synchronize()
t = buf.cpu().double()
names_h = ["start", "dma issued", "pair0 mfma", "pairs 1,2", "last gelu", "waits", "barrier"]
names_p = ["start", "-", "stage C loop", "stores", "stage A", "lgkm wait", "barrier"]
for w in range(8):
    tw = t[w]
    ok = (tw[:, 0] > 0) & (tw[:, 6] > 0)
    tw = tw[ok]
    if len(tw) == 0:
        print(f"wave {w}: no stamps")
        continue
    names = names_h if w >= 4 else names_p
    pts = [0, 1, 2, 3, 4, 5, 6] if w >= 4 else [0, 2, 3, 4, 5, 6]
    segs = []
    for a, b in zip(pts[:-1], pts[1:]):
        d = (tw[:, b] - tw[:, a]).mean().item()
        segs.append(f"{names[b]} {d:7.0f}")
    trip = (tw[1:, 0] - tw[:-1, 0]).mean().item()
    print(f"wave {w} ({'H' if w >= 4 else 'P'}) trip {trip:7.0f} ticks | " + " | ".join(segs))
