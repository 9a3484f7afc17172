#!/usr/bin/env python3
"""Small fixed workloads through every engine, outputs saved to argv[1] (a .pt file): the A/B switches of the library
(``NUNIF_*`` read once per process in ``nunif_amd/csrc``) are exercised by running this under two environments and comparing the
files (tests/test_ab_switches.py).  Workloads: swin_unet 2x and 1x tiled renders (stem, PatchUp / PatchDown, tails, stitcher),
cunet and upcunet tile batches (stem, down / up kernels, sliced bottom convs, SE fusion, image heads), the Depth-Anything ViT-S
(out_conv order, RCU1 branch), light_inpaint_v1 (conv slices), the forward warp (its two instruction streams)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import synthetic as S  # noqa: E402
from nunif_amd.nunif.models import create_model  # noqa: E402
from nunif_amd.nunif.utils.render import tiled_render  # noqa: E402
import nunif_amd.waifu2x.utils  # noqa: E402,F401

torch.set_grad_enabled(False)
dev = "cuda:0"
g = torch.Generator().manual_seed(77)
img = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 300, 400, generator=g), 3, stride=1, padding=1)[0].to(dev)
out = {}
for name, scale in (("waifu2x.swin_unet_2x", 2), ("waifu2x.swin_unet_1x", 1)):
    m = create_model(name).eval()
    m.load_state_dict(S.swin_unet_state_dict(11, scale_factor=scale))
    m = m.to(dev)
    out[name] = tiled_render(img, m, tile_size=256, batch_size=4).float().cpu()
tiles = torch.stack([img[:, :256, :256], img[:, 44:, 100:356]])
for name, up in (("waifu2x.cunet", False), ("waifu2x.upcunet", True)):
    m = create_model(name).eval()
    m.load_state_dict(S.cunet_state_dict(7, up=up))
    m = m.to(dev)
    out[name] = m(tiles).float().cpu()
    # more than NUNIF_CONV3_DMA_MIN (24) patches per launch as well: a 12-tile minibatch
    out[name + ".12"] = m(tiles.repeat(6, 1, 1, 1)).float().cpu()[:2]
from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2  # noqa: E402
net = HipDepthAnythingV2(S.depth_anything_v2_state_dict(601), dev)
xd = torch.nn.functional.interpolate(tiles, size=(126, 154), mode="bilinear") * 2 - 1
out["depth_vits"] = net(xd).float().cpu()
from nunif_amd.iw3.models.light_inpaint_v1 import LightInpaintV1  # noqa: E402
li = LightInpaintV1().eval()
li.load_state_dict(S.light_inpaint_state_dict(701))
li = li.to(dev)
mask = torch.zeros(2, 1, 256, 256, device=dev, dtype=torch.bool)
mask[:, :, 60:200, 90:120] = True
out["light_inpaint_v1"] = li.infer(tiles, mask).float().cpu()
from nunif_amd.iw3.forward_warp import apply_divergence_forward_warp  # noqa: E402
d = S.synth_depth(5, 2, 256, 256, "smooth_edges").to(dev)
le, ri = apply_divergence_forward_warp(tiles, d, 4.0, 0.5, method="forward_fill", synthetic_view="both")
out["forward_fill"] = torch.cat([le, ri], dim=1).float().cpu()
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
print("OK", {k: tuple(v.shape) for k, v in out.items()})
