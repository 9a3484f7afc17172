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


from .base_depth_model import BaseDepthModel  # noqa: E402


class VideoDepthAnythingModel(BaseDepthModel):
    """``VideoDepthAnythingModel`` (reference :110-283): the NON-streaming wrapper.  The external online model buffers frames —
    ``model.infer(frame | None, use_amp)`` returns ``None`` or a list of finished depth maps, possibly later than the frame
    that completed them — so the wrapper counts frames in and out, normalises what comes back through the EMA scaler and drains
    the model (``infer(None)``) at scene cuts / end of stream, dropping the padding the model emits past the last real frame.

    The network itself lives in the external ``torch.hub`` repository (``VideoDepthAnythingOnline`` / ``Metric...Online``,
    :127-146) and is not restated: ``load_model`` takes the ``backbone`` given to the constructor — any object with
    ``infer(frame | None, use_amp=...)``, ``reset_state()`` and a ``metric_depth`` attribute.  Pre / post-processing are the
    HIP paths above."""

    def __init__(self, model_type, backbone=None, depth_aa=None):
        super().__init__(model_type)
        if model_type not in NAME_MAP:
            raise ValueError(f"unknown model_type {model_type}")
        self.input_frame_count = 0
        self.output_frame_count = 0
        self.metric_depth = model_type in METRIC_DEPTH_TYPES
        self.force_disparity = True
        self._backbone = backbone
        self.depth_aa = depth_aa

    def load_model(self, model_type, resolution=None, device=None, backbone=None, **kwargs):
        model = backbone if backbone is not None else self._backbone
        if model is None or not hasattr(model, "infer"):
            raise RuntimeError("VideoDepthAnythingModel: the online network lives in an external torch.hub repository; pass "
                               "backbone=<object with infer(frame | None, use_amp) / reset_state / metric_depth>")
        if not hasattr(model, "metric_depth"):
            model.metric_depth = self.metric_depth
        model.prep_lower_bound = resolution or 392
        if model.prep_lower_bound % 14 != 0:
            model.prep_lower_bound += 14 - model.prep_lower_bound % 14
        return model

    def reset_state(self):
        self.model.reset_state()
        self.input_frame_count = 0
        self.output_frame_count = 0

    def reset(self):
        self.reset_state()
        self.reset_ema()

    def _post(self, frames, edge_dilation, depth_aa, enable_amp):
        return postprocess(torch.stack(frames), edge_dilation=edge_dilation, depth_aa=depth_aa,
                           metric_depth=self.model.metric_depth, force_disparity=self.force_disparity, enable_amp=enable_amp)

    @torch.inference_mode()
    def infer(self, x, enable_amp=True, edge_dilation=0, **kwargs):
        """Single image through the video model (reference :166-190, marked "DONT USE THIS" there): push one frame, drain."""
        if not torch.is_tensor(x):
            raise ValueError("infer expects a CHW or BCHW float tensor in [0,1]")
        batch = x.ndim != 3
        if not batch:
            x = x.unsqueeze(0)
        self.reset()
        x = batch_preprocess(x.to(self.device), self.model.prep_lower_bound, metric_depth=self.model.metric_depth,
                             limit_resolution=self.limit_resolution)
        self.model.infer(x[0], use_amp=enable_amp)
        self.input_frame_count = 1
        out = self._post(self._flush(), edge_dilation, None, enable_amp)
        self.reset()
        return out if batch else out.squeeze(0)

    @torch.inference_mode()
    def infer_with_normalize(self, x, pts, reset_pts, enable_amp=True, edge_dilation=0, depth_aa=None, **kwargs):
        assert x.ndim == 4
        depth_aa = self.depth_aa if depth_aa else None
        x = batch_preprocess(x.to(self.device), self.model.prep_lower_bound, metric_depth=self.model.metric_depth,
                             limit_resolution=self.limit_resolution)
        outputs = []
        for i in range(x.shape[0]):
            self.input_frame_count += 1
            ret = self.model.infer(x[i], use_amp=enable_amp)
            if ret is not None:
                self.output_frame_count += len(ret)
                out = self._post(ret, edge_dilation, depth_aa, enable_amp)
                for j in range(out.shape[0]):
                    normalized_depth = self.minmax_normalize_chw(out[j])
                    if normalized_depth is not None:
                        outputs.append(normalized_depth)
            if pts[i] in reset_pts:
                outputs += self.flush_with_normalize(enable_amp=enable_amp, edge_dilation=edge_dilation, depth_aa=depth_aa)
                self.reset()
        return outputs

    @torch.inference_mode()
    def flush_with_normalize(self, enable_amp=True, edge_dilation=0, depth_aa=None):
        if isinstance(depth_aa, bool):
            depth_aa = self.depth_aa if depth_aa else None
        outputs = []
        ret = self._flush(enable_amp=enable_amp)
        if ret:
            out = self._post(ret, edge_dilation, depth_aa, enable_amp)
            for i in range(out.shape[0]):
                normalized_depth = self.minmax_normalize_chw(out[i])
                if normalized_depth is not None:
                    outputs.append(normalized_depth)
            outputs += self.flush_minmax_normalize()
        return outputs

    def _flush(self, enable_amp=True):
        results = []
        while self.output_frame_count < self.input_frame_count:
            ret = self.model.infer(None, use_amp=enable_amp)
            if ret is None:
                continue
            results += ret
            self.output_frame_count += len(ret)
        if results:
            unpad = self.output_frame_count - self.input_frame_count
            if unpad > 0:
                assert unpad <= len(results)
                results = results[:-unpad]
            return results
        return []

    @classmethod
    def get_name(cls):
        return "VideoDepthAnything"

    def is_image_supported(self):
        return False

    @classmethod
    def supported(cls, model_type):
        return model_type in NAME_MAP

    def is_metric(self):
        if not self.metric_depth:
            return False
        return not self.force_disparity

    @classmethod
    def multi_gpu_supported(cls, model_type):
        return False
