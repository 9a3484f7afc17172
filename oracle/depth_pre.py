"""Oracle: iw3 depth-model pre/post-processing around the (external) depth network (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Follows ``iw3/depth_anything_model.py`` ``batch_preprocess`` :69-110 and ``iw3/depth_scaler.py`` ``minmax_normalize``
:4-17.  The resampling itself is ATen's ``F.interpolate`` — torch-CPU is the oracle for it.
"""
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
MIN_RESOLUTION = 224


def preprocess_size(h, w, lower_bound=392, max_aspect_ratio=4, limit_resolution=False, multiple=14):
    """Integer target size of batch_preprocess (:73-99)."""
    if limit_resolution and lower_bound > min(w, h):
        lower_bound = min(w, h)
        lower_bound -= lower_bound % multiple
        lower_bound = max(lower_bound, MIN_RESOLUTION)
    s = lower_bound / w if w < h else lower_bound / h
    nh, nw = int(h * s), int(w * s)
    if nh < nw:
        nw = min(nw, int(max_aspect_ratio * nh))
    else:
        nh = min(nh, int(max_aspect_ratio * nw))
    nh -= nh % multiple
    nw -= nw % multiple
    return max(nh, lower_bound), max(nw, lower_bound)


def batch_preprocess(x, lower_bound=392, max_aspect_ratio=4, limit_resolution=False):
    nh, nw = preprocess_size(x.shape[2], x.shape[3], lower_bound, max_aspect_ratio, limit_resolution)
    x = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True).clamp(0, 1)
    mean = torch.tensor(MEAN).view(1, 3, 1, 1)
    std = torch.tensor(STD).view(1, 3, 1, 1)
    return (x - mean) / std


def minmax_normalize(frame):
    """minmax_normalize with the frame's own min/max (EMA off: decay 0, buffer 1 — base_depth_model.py:39-41)."""
    lo, hi = frame.amin(), frame.amax()
    scale = hi - lo
    if scale > 0:
        return ((frame - lo) / scale).clamp(0.0, 1.0)
    return frame.clamp(0.0, 1.0)
