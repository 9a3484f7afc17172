"""Pre/post-processing of the VideoDepthAnything wrappers on the HIP engine.

Mirrors the in-tree functions of ``iw3/video_depth_anything_model.py``: ``batch_preprocess`` :51-58 (the Depth-Anything
resize/normalise at ``lower_bound - 28`` followed by a 14-px reflection pad for the metric checkpoints) and ``_postprocess`` /
``postprocess`` :61-107 (nan_to_num, max-distance clamp, metric depth -> disparity ``1/(d+0.1)``, DepthAA, crop of the
reflection pad, edge dilation, sign convention).  The network between them is external to the reference tree (``torch.hub``
repository ``nagadomi/Video-Depth-Anything_iw3``, :133-145); see ``video_depth_anything_streaming_model.py``.
"""
import torch

from . import _ops
from .depth_anything_model import batch_preprocess as batch_preprocess_da
from .dilation import dilate_edge, edge_dilation_is_enabled

METRIC_PADDING = 14
NAME_MAP = {
    "VDA_S": "vits", "VDA_B": "vitb", "VDA_L": "vitl", "VDA_Metric": "vitl",
    "VDA_Metric_S": "vits", "VDA_Metric_B": "vitb", "VDA_Metric_L": "vitl",
}
METRIC_DEPTH_TYPES = {"VDA_Metric", "VDA_Metric_S", "VDA_Metric_B", "VDA_Metric_L"}


def batch_preprocess(x, lower_bound, metric_depth, limit_resolution=False):
    if metric_depth:
        x = batch_preprocess_da(x, lower_bound - METRIC_PADDING * 2, limit_resolution=limit_resolution)
        x = _ops.reflection_pad2d(x, (METRIC_PADDING,) * 4)
    else:
        x = batch_preprocess_da(x, lower_bound, limit_resolution=limit_resolution)
    assert x.shape[2] % 14 == 0 and x.shape[3] % 14 == 0
    return x


def _postprocess(out, edge_dilation, metric_depth, force_disparity=False, max_dist=None, depth_aa=None, enable_amp=True):
    out = out.unsqueeze(1)
    to_disp = bool(metric_depth and force_disparity)
    is_disparity = (not metric_depth) or to_disp
    out = _ops.depth_postprocess(out, max_dist=max_dist, to_disparity=to_disp, eps=0.1)
    if depth_aa is not None:
        out = depth_aa.infer(out)
    if metric_depth:
        out = _ops.reflection_pad2d(out, (-METRIC_PADDING,) * 4)             # F.pad(out, (-14,) * 4)
    if edge_dilation_is_enabled(edge_dilation):
        if is_disparity:
            out = dilate_edge(out, edge_dilation)
        else:
            out = -dilate_edge(-out, edge_dilation)
    if not is_disparity:
        out = -out                                                             # zoedepth-compatible sign (:88-90)
    return out.float()


def postprocess(out, edge_dilation, metric_depth, max_dist=None, depth_aa=None, force_disparity=False, enable_amp=True):
    micro_batch_size = 4                                                       # :96 (bounds DepthAA's working set)
    return torch.cat([
        _postprocess(batch, edge_dilation=edge_dilation, metric_depth=metric_depth, force_disparity=force_disparity,
                     max_dist=max_dist, depth_aa=depth_aa, enable_amp=enable_amp)
        for batch in torch.split(out, micro_batch_size, dim=0)], dim=0)
