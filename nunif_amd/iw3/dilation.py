"""Edge-weighted depth dilation on the HIP engine.  Mirrors ``iw3/dilation.py``: ``edge_dilation_parse`` :5-22,
``edge_dilation_is_enabled`` :25-27, ``dilate_edge`` :116-142 (edge_weight / gaussian_blur / dilate are fused into
``nunif_hip_dilate_edge``, nunif_amd/csrc/iw3_depth.hip)."""
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
