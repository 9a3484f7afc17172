"""Oracle: iw3 edge-weighted depth dilation (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Follows ``iw3/dilation.py`` (reference): ``edge_dilation_parse`` :5-22, ``gaussian_blur`` :30-38, ``dilate`` :41-46,
``edge_weight`` :101-113, ``dilate_edge`` :116-142.
"""
import torch
import torch.nn.functional as F

GAUSS = torch.tensor([[21.0, 31.0, 21.0], [31.0, 48.0, 31.0], [21.0, 31.0, 21.0]]) / 256.0


def parse(n):
    """(x_iter, y_iter) from int | list | tuple | None."""
    if n is None:
        return 0, 0
    if isinstance(n, int):
        return n, n
    if isinstance(n, (list, tuple)):
        if len(n) == 0:
            return 0, 0
        return (n[0], n[0]) if len(n) == 1 else (n[0], n[1])
    raise ValueError(f"Unsupported edge_dilation type {type(n)}")


def local_range_weight(x):
    """3x3 (max - min), z-scored per image, clamped to +-3, min-max normalised to [0,1]."""
    hi = F.max_pool2d(x, 3, stride=1, padding=1)
    lo = -F.max_pool2d(-x, 3, stride=1, padding=1)
    r = hi - lo
    rc = r - r.mean(dim=(1, 2, 3), keepdim=True)
    rs = rc.pow(2).mean(dim=(1, 2, 3), keepdim=True).sqrt()
    w = (rc / (rs + 1e-6)).clamp(-3, 3)
    lo_w, hi_w = w.amin(dim=(1, 2, 3), keepdim=True), w.amax(dim=(1, 2, 3), keepdim=True)
    return (w - lo_w) / ((hi_w - lo_w) + 1e-6)


def one_iteration(x, ky, kx):
    w = local_range_weight(x)
    blur = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), GAUSS.view(1, 1, 3, 3))
    x2 = F.max_pool2d(blur, kernel_size=(ky, kx), stride=1, padding=(ky // 2, kx // 2))
    return x * (1 - w) + x2 * w


def dilate_edge(x, n):
    nx, ny = parse(n)
    both = min(nx, ny)
    for _ in range(both):
        x = one_iteration(x, 3, 3)
    for _ in range(ny - both):
        x = one_iteration(x, 3, 1)
    for _ in range(nx - both):
        x = one_iteration(x, 1, 3)
    return x
