"""Pre/post-processing of the Depth-Anything wrapper on the HIP engine.

Mirrors the in-tree parts of ``iw3/depth_anything_model.py``: ``batch_preprocess`` :69-110 (size rule + antialiased
bilinear resize + clamp + ImageNet normalise in ONE kernel pair) and the post-network steps of ``batch_infer``
:123-182 that do not involve the network (edge dilation, flip merge).  The DINOv2/DPT backbone itself is external
to the reference tree (``torch.hub`` repo, :200-230) and is a "next" row: ``batch_infer`` takes
any callable ``model(x[B,3,h,w]) -> [B,h,w]``.
"""
import torch

from . import _ops
from .dilation import dilate_edge, edge_dilation_is_enabled

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
MIN_RESOLUTION = 224


def preprocess_size(H, W, lower_bound=392, max_aspect_ratio=4, limit_resolution=False):
    multiple = 14
    if limit_resolution and lower_bound > min(W, H):
        lower_bound = min(W, H)
        lower_bound -= lower_bound % multiple
        lower_bound = max(lower_bound, MIN_RESOLUTION)
    scale = lower_bound / W if W < H else lower_bound / H
    new_h, new_w = int(H * scale), int(W * scale)
    if new_h < new_w:
        new_w = min(new_w, int(max_aspect_ratio * new_h))
    else:
        new_h = min(new_h, int(max_aspect_ratio * new_w))
    new_h -= new_h % multiple
    new_w -= new_w % multiple
    return max(new_h, lower_bound), max(new_w, lower_bound)


def batch_preprocess(x, lower_bound=392, max_aspect_ratio=4, limit_resolution=False):
    size = preprocess_size(x.shape[2], x.shape[3], lower_bound, max_aspect_ratio, limit_resolution)
    return _ops.resize_aa(x, size, mode="bilinear", align_corners=False, clamp01=True, mean=MEAN, std=STD)


@torch.inference_mode()
def batch_infer(model, im, flip_aug=True, enable_amp=False, edge_dilation=2, lower_bound=392,
                limit_resolution=False, metric_depth=False, depth_aa=None, **_):
    single = im.dim() == 3
    x = batch_preprocess(im.unsqueeze(0) if single else im, lower_bound, limit_resolution=limit_resolution)
    if flip_aug:
        x = torch.cat([x, torch.flip(x, dims=[3])], dim=0)
    out = model(x).unsqueeze(1).float()
    # torch.nan_to_num (reference :150): the engine's depth_post_kernel with every other step switched off
    out = _ops.depth_postprocess(out) if out.is_cuda else torch.nan_to_num(out)
    if depth_aa is not None:
        out = depth_aa.infer(out)                      # depth_anything_model.py:153-154 (nunif_amd.iw3.models.DepthAA)
    if edge_dilation_is_enabled(edge_dilation):
        out = dilate_edge(-out if metric_depth else out, edge_dilation)          # :156-160: metric = -dilate_edge(-out)
        out = -out if metric_depth else out
    if metric_depth:
        out = -out                                                             # :162-164 "invert for zoedepth compatibility"
    if flip_aug:
        a, b = out.chunk(2, dim=0)
        out = (a + torch.flip(b, dims=[3])) * 0.5
    return out[0] if single else out
