"""Oracle: iw3 grid-sample backward warp (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Follows ``iw3/backward_warp.py`` (reference): ``backward_warp`` :67-83, ``make_grid`` :86-93,
``apply_divergence_grid_sample`` :96-121.
"""
import torch
import torch.nn.functional as F


def identity_grid(b, h, w):
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    return torch.stack([gx, gy], 0).unsqueeze(0).expand(b, 2, h, w)


def sample(c, grid):
    if c.shape[2:] != grid.shape[2:]:
        grid = F.interpolate(grid, size=c.shape[-2:], mode="bilinear", align_corners=True, antialias=False)
    z = F.grid_sample(c, grid.permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=True)
    return torch.clamp(z, 0, 1)


def grid_sample_warp(c, depth, divergence, convergence, synthetic_view="both"):
    assert synthetic_view in ("both", "left", "right")
    b, _, h, w = depth.shape
    if synthetic_view != "both":
        divergence = divergence * 2
    shift_size = divergence * 0.01
    shift = depth * shift_size - (shift_size * convergence)
    delta = torch.cat([shift, torch.zeros_like(shift)], 1)
    scale = max(h, w) / w
    grid = identity_grid(b, h, w)
    left = sample(c, grid + (-delta) * scale) if synthetic_view in ("both", "left") else c
    right = sample(c, grid + delta * scale) if synthetic_view in ("both", "right") else c
    return left, right
