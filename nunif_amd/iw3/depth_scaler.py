"""Depth normalisation.  Mirrors ``iw3/depth_scaler.py``: ``minmax_normalize`` :4-17 and the default (EMA off)
behaviour of ``EMAMinMaxScaler`` (decay 0 / buffer 1, ``iw3/base_depth_model.py:39-41``): each frame is scaled by its
own min/max.  The look-ahead EMA buffer (:64-142) is sequential host logic and a "next" row."""
import torch

from . import _ops


def minmax_normalize(frame, min_value=None, max_value=None):
    if min_value is None and max_value is None:
        single = frame.dim() == 3
        y = _ops.minmax_normalize(frame.unsqueeze(0) if single else frame)
        return y[0] if single else y
    scale = max_value - min_value
    if scale > 0:
        return ((frame - min_value) / scale).clamp(0.0, 1.0)
    return frame.clamp(0.0, 1.0)


def minmax_normalize_chw(depth):
    return minmax_normalize(depth)
