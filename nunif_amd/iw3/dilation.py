"""Edge-weighted depth dilation on the HIP engine.  Mirrors ``iw3/dilation.py``: ``edge_dilation_parse`` :5-22,
``edge_dilation_is_enabled`` :25-27, ``dilate_edge`` :116-142 (edge_weight / gaussian_blur / dilate are fused into
``nunif_hip_dilate_edge``, nunif_amd/csrc/iw3_depth.hip), and the mask morphology helpers ``dilate`` :41-46, ``erode`` :49-54,
``closing`` :57-64, ``mask_closing`` :145-153, ``dilate_outer`` :67-81, ``dilate_inner`` :84-98 (``nunif_hip_mask_morphology``;
3x3 kernels only, as every call site in the reference uses)."""
import torch

from . import _ops


def edge_dilation_parse(edge_dilation):
    if isinstance(edge_dilation, (list, tuple)):
        if len(edge_dilation) == 0:
            return 0, 0
        if len(edge_dilation) == 1:
            return edge_dilation[0], edge_dilation[0]
        return edge_dilation[0], edge_dilation[1]
    if isinstance(edge_dilation, int):
        return edge_dilation, edge_dilation
    if edge_dilation is None:
        return 0, 0
    raise ValueError(f"Unsupported edge_dilation type {type(edge_dilation)}. Supported types: int, list, tuple.")


def edge_dilation_is_enabled(edge_dilation):
    x, y = edge_dilation_parse(edge_dilation)
    return x != 0 or y != 0


def dilate_edge(x, n):
    x_iter, y_iter = edge_dilation_parse(n)
    return _ops.dilate_edge(x, x_iter, y_iter).to(x.dtype)


def _k3(kernel_size):
    if kernel_size not in (3, (3, 3), [3, 3]):
        raise NotImplementedError("the HIP mask morphology implements the 3x3 kernel the reference's call sites use")


def dilate(mask, kernel_size=3):
    _k3(kernel_size)
    return _ops.mask_morphology(mask, 0, 1).to(mask.dtype if mask.is_floating_point() else torch.float32)


def erode(mask, kernel_size=3):
    _k3(kernel_size)
    return _ops.mask_morphology(mask, 1, 1).to(mask.dtype if mask.is_floating_point() else torch.float32)


def closing(mask, kernel_size=3, n_iter=2):
    _k3(kernel_size)
    return _ops.mask_morphology(mask, 2, n_iter)


def mask_closing(mask, kernel_size=3, n_iter=2):
    _k3(kernel_size)
    return _ops.mask_morphology(mask, 3, n_iter)


def _scaled(mask, n_iter, base_width):
    return max(round(mask.shape[-1] / base_width * n_iter), 1) if base_width is not None else n_iter


def dilate_outer(mask, n_iter, base_width=None):
    if n_iter <= 0:
        return mask
    return _ops.mask_morphology(mask, 4, 0, _scaled(mask, n_iter, base_width)).to(mask.dtype)


def dilate_inner(mask, n_iter, base_width=None):
    if n_iter <= 0:
        return mask
    return _ops.mask_morphology(mask, 4, _scaled(mask, n_iter, base_width), 0).to(mask.dtype)
