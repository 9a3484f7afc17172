"""Bleed RGB into the transparent border before super-resolution.  Mirrors ``nunif/utils/alpha.py``
``AlphaBorderPadding`` :32-57 (``ChannelWiseSum`` :5-29 is the 3x3 box sum).  Runs as ATen ops on the ROCm device:
only images with a non-trivial alpha channel reach it (waifu2x/utils.py:266-271); a fused HIP kernel is a "next" row
(SURVEY.md §8f f4)."""
import torch
import torch.nn.functional as F


def _box3(x):
    return F.avg_pool2d(x.unsqueeze(0), 3, stride=1, padding=1, divisor_override=1)[0]


class AlphaBorderPadding(torch.nn.Module):
    def forward(self, rgb, alpha, offset):
        assert rgb.ndim == 3 and alpha.ndim == 3 and rgb.shape[0] == 3 and alpha.shape[0] == 1
        rgb = rgb.clone()
        mask = (alpha > 0).to(rgb.dtype)              # [1,H,W]
        hole = mask < 1.0
        rgb = torch.where(hole, torch.zeros((), dtype=rgb.dtype, device=rgb.device), rgb)
        for _ in range(offset):
            weight = _box3(mask)
            border = _box3(rgb) / (weight + 1e-7)
            rgb = torch.where(hole, border, rgb)
            mask = (weight > 0).to(rgb.dtype)
            hole = mask < 1.0
        return rgb.clamp_(0.0, 1.0)
