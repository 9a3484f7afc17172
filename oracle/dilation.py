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


# ---- mask morphology (iw3/dilation.py dilate :41-46, erode :49-54, closing :57-64, mask_closing :145-153,
#      dilate_outer :67-81, dilate_inner :84-98) -------------------------------------------------------------------------
def dilate(mask, kernel_size=3):
    return F.max_pool2d(mask, kernel_size=kernel_size, stride=1, padding=kernel_size // 2)


def erode(mask, kernel_size=3):
    return -F.max_pool2d(-mask, kernel_size=kernel_size, stride=1, padding=kernel_size // 2)


def closing(mask, kernel_size=3, n_iter=2):
    mask = mask.float()
    for _ in range(n_iter):
        mask = dilate(mask, kernel_size)
    for _ in range(n_iter):
        mask = erode(mask, kernel_size)
    return mask


def mask_closing(mask, kernel_size=3, n_iter=2):
    org = mask.float()
    return (closing(org, kernel_size, n_iter) + org).clamp(0, 1)


def dilate_outer(mask, n_iter, base_width=None):
    if n_iter <= 0:
        return mask
    dt, m = mask.dtype, mask.bool()
    if base_width is not None:
        n_iter = max(round(mask.shape[-1] / base_width * n_iter), 1)
    for _ in range(n_iter):
        m = m | F.pad(m, (1, 0, 0, 0))[:, :, :, :-1]
    return m.to(dt)


def dilate_inner(mask, n_iter, base_width=None):
    if n_iter <= 0:
        return mask
    dt, m = mask.dtype, mask.bool()
    if base_width is not None:
        n_iter = max(round(mask.shape[-1] / base_width * n_iter), 1)
    for _ in range(n_iter):
        m = m | F.pad(m, (0, 1, 0, 0))[:, :, :, 1:]
    return m.to(dt)
