#!/usr/bin/env python3
"""Fixtures produced by INDEPENDENT implementations (HuggingFace ``transformers``) of the two external networks on the hot
path, so that the GPU tests compare the HIP kernels with something our own restatements had no hand in:

    python tests/golden/make_golden_hf.py [swin] [depth]

``swin_unet_hf.npz``  — the REFERENCE's ``SwinUNet`` / ``SwinUNet2x`` / ``SwinUNet4x`` (``waifu2x/models/swin_unet.py``,
    imported from /root/reference) running over ``oracle.hf_pin.HFSwinTransformerBlock``: torchvision's constructor and
    state-dict layout, HuggingFace ``SwinLayer`` arithmetic.  Same seeded weights and inputs as ``swin_unet.npz``.
``depth_anything_hf.npz`` — ``transformers.DepthAnythingForDepthEstimation`` (``Dinov2Backbone`` + DPT neck / head) loaded
    with the seeded Depth-Anything-V2 ViT-S checkpoint-layout weights (``nunif_amd.synthetic``), with the upstream
    position-embedding interpolation (``oracle/hf_pin.py`` documents that one divergence); a 2 x 56 x 70 batch and the
    392 x 686 map that ``batch_preprocess`` makes of a 1080p frame.

Build container only (needs ``transformers`` and, for ``swin``, /root/reference).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.set_grad_enabled(False)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def sd_checksum(sd):
    return float(sum(v.double().sum().item() for v in sd.values() if v.is_floating_point()))


def gen_swin():
    import transformers.models.swin.modeling_swin  # noqa: F401   (before the torchvision stub goes in: transformers probes it)
    from oracle import refstub
    refstub.install(swin_block="hf")
    from waifu2x.models.swin_unet import SwinUNet, SwinUNet2x, SwinUNet4x
    from oracle import hf_pin
    from oracle import swin_unet as O
    assert sys.modules["torchvision.models.swin_transformer"].SwinTransformerBlock is hf_pin.HFSwinTransformerBlock
    old = np.load(os.path.join(HERE, "swin_unet.npz"))
    x = torch.from_numpy(old["x"])
    out = {"x": x}
    for cls, sf, tag in ((SwinUNet, 1, "1x"), (SwinUNet2x, 2, "2x"), (SwinUNet4x, 4, "4x")):
        sd = O.random_state_dict(100 + sf, sf)
        m = cls().eval()
        m.load_state_dict(sd, strict=True)
        y = m(x)
        out["y_" + tag] = y
        out["sdsum_" + tag] = sd_checksum(sd)
        d = (y - torch.from_numpy(old["y_" + tag])).abs().max().item()
        print(f"swin {tag}: reference U-Net over the HF block vs over oracle.tv_swin_block: max |diff| = {d:.3g}")
        assert d < 2e-4, d
    x2 = torch.from_numpy(old["x_112"])
    sd = O.random_state_dict(102, 2)
    m = SwinUNet2x().eval()
    m.load_state_dict(sd, strict=True)
    out["x_112"] = x2
    out["y_2x_112"] = m(x2)
    save("swin_unet_hf", **out)


def gen_depth():
    from conftest import synth_image
    from oracle import hf_pin
    from nunif_amd.synthetic import depth_anything_v2_state_dict
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    out = {}
    sd = depth_anything_v2_state_dict(601)
    out["sdsum"] = sd_checksum(sd)
    x = (torch.stack([synth_image(700 + i, 3, 56, 70) for i in range(2)]) - mean) / std
    out["x_small"] = x
    out["y_small"] = hf_pin.depth_anything_hf_forward(sd, x)
    x = (synth_image(95, 3, 392, 686)[None] - mean) / std          # seed 95 = tests/test_depth_anything.py's full-size case
    out["y_392x686_seed95"] = hf_pin.depth_anything_hf_forward(sd, x)
    # metric head (Sigmoid x max_depth) and Depth-Anything V1's taps on the small geometry
    sd8 = depth_anything_v2_state_dict(620, grid=8, encoder="vits")
    out["sdsum8"] = sd_checksum(sd8)
    x = (torch.stack([synth_image(190 + i, 3, 70, 98) for i in range(2)]) - mean) / std
    out["y_metric80"] = hf_pin.depth_anything_hf_forward(sd8, x, max_depth=80.0)
    out["y_v1taps"] = hf_pin.depth_anything_hf_forward(sd8, x, taps=(8, 9, 10, 11))
    save("depth_anything_hf", **out)


GROUPS = {"swin": gen_swin, "depth": gen_depth}

if __name__ == "__main__":
    for g in (sys.argv[1:] or list(GROUPS)):
        GROUPS[g]()
