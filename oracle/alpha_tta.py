"""Oracle: alpha border padding, 8-way TTA and the Waifu2x.convert glue (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Follows ``nunif/utils/alpha.py`` ``AlphaBorderPadding`` :32-57 (``ChannelWiseSum`` :5-29), ``nunif/transforms/tta.py``
``tta_split`` :20-34 / ``tta_merge`` :37-48 and ``waifu2x/utils.py`` ``Waifu2x.convert`` :255-297.
"""
import torch
import torch.nn.functional as F


def box_sum3(x):
    """3x3 box sum with zero padding, per channel (== depthwise conv with an all-ones kernel)."""
    c = x.shape[0]
    return F.conv2d(x.unsqueeze(0), torch.ones(c, 1, 3, 3), padding=1, groups=c)[0]


def alpha_border_padding(rgb, alpha, offset):
    rgb = rgb.clone()
    a = alpha[0]
    mask = (a > 0).float()
    hole = mask < 1.0
    rgb[:, hole] = 0.0
    for _ in range(offset):
        weight = box_sum3(mask.unsqueeze(0))[0]
        border = box_sum3(rgb) / (weight + 1e-7)
        rgb[:, hole] = border[:, hole]
        mask = (weight > 0).float()
        hole = mask < 1.0
    return rgb.clamp(0.0, 1.0)


def tta_split(x):
    tr = torch.rot90(x, 1, (1, 2))
    views = []
    for base in (x, tr):
        v = torch.flip(base, (1,))
        views += [base, torch.flip(base, (2,)), v, torch.flip(v, (2,))]
    return tuple(views)


def tta_merge(xs):
    out = torch.zeros_like(xs[0])
    for k, y in enumerate(xs):
        if k & 1:
            y = torch.flip(y, (2,))
        if k & 2:
            y = torch.flip(y, (1,))
        if k & 4:
            y = torch.rot90(y, -1, (1, 2))
        out = out + y
    return torch.clamp(out * (1 / 8.0), 0, 1)


def convert(render, x, alpha, scale, offset, tta=False):
    """Waifu2x.convert for a scale method whose colour model is also the alpha model (utils.py:255-297).
    ``render(chw) -> chw`` is the tiled render of that model."""
    blank = alpha is None or bool(torch.all(alpha == 1))
    if alpha is not None and not blank:
        x = alpha_border_padding(x, alpha, offset)
    rgb = tta_merge([render(v) for v in tta_split(x)]) if tta else render(x)
    if alpha is not None:
        if blank:
            alpha = F.interpolate(alpha.unsqueeze(0), scale_factor=scale, mode="nearest").squeeze(0)
        else:
            alpha = render(alpha.expand(3, -1, -1)).mean(0, keepdim=True)
    return rgb, alpha
