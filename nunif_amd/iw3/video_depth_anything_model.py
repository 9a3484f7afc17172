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


class _OnlineLedger:
    """Book-keeping around an ONLINE video depth network: ``net.infer(frame | None, use_amp)`` swallows frames and hands back
    lists of finished depth maps whenever it likes (``None`` while it is buffering), and once it is drained with ``None`` it
    keeps emitting padding.  The ledger counts what went in and what came out; ``feed`` pushes one frame, ``drain`` pulls until
    every frame that went in has come back and cuts the padding past the last real frame off.  (The reference keeps two
    counters on the wrapper and repeats this arithmetic in ``infer`` / ``_flush``, video_depth_anything_model.py:166-283.)"""

    def __init__(self, net):
        self.net = net
        self.n_in = self.n_out = 0

    def owed(self):
        return self.n_in - self.n_out

    def feed(self, frame, use_amp):
        self.n_in += 1
        got = self.net.infer(frame, use_amp=use_amp) or []
        self.n_out += len(got)
        return got

    def drain(self, use_amp):
        got = []
        while self.owed() > 0:
            more = self.net.infer(None, use_amp=use_amp)
            if more:
                got += more
                self.n_out += len(more)
        surplus = -self.owed()                     # padding frames the network emitted past the last real one
        assert 0 <= surplus <= len(got) or not got
        return got[:len(got) - surplus] if surplus > 0 else got

    def restart(self):
        self.net.reset_state()
        self.n_in = self.n_out = 0


class VideoDepthAnythingModel(BaseDepthModel):
    """The NON-streaming wrapper (reference :110-283) around an external online network (``VideoDepthAnythingOnline`` /
    ``Metric...Online`` of the ``torch.hub`` repository, :127-146 — not restated; ``load_model`` takes the ``backbone`` given to
    the constructor: any object with ``infer(frame | None, use_amp=...)``, ``reset_state()`` and ``metric_depth``).

    Frames go in one by one (:class:`_OnlineLedger`); whatever comes back is post-processed by the HIP paths above and pushed
    through the EMA min-max scaler; at a scene cut / end of stream the network is drained, the scaler flushed, everything
    restarted.  Contract pinned by ``tests/golden/video_depth_anything_online.npz`` (the reference class around a fake net)."""

    def __init__(self, model_type, backbone=None, depth_aa=None):
        super().__init__(model_type)
        if model_type not in NAME_MAP:
            raise ValueError(f"unknown model_type {model_type}")
        self.metric_depth = model_type in METRIC_DEPTH_TYPES
        self.force_disparity = True             # 1 / depth for the metric checkpoints; is_metric() is then False
        self._backbone, self.depth_aa = backbone, depth_aa
        self._ledger = None

    # the reference's two counters, for callers that peek at them
    input_frame_count = property(lambda self: self._ledger.n_in if self._ledger else 0)
    output_frame_count = property(lambda self: self._ledger.n_out if self._ledger else 0)

    def load_model(self, model_type, resolution=None, device=None, backbone=None, **kwargs):
        net = backbone if backbone is not None else self._backbone
        if net is None or not hasattr(net, "infer"):
            raise RuntimeError("VideoDepthAnythingModel: the online network lives in an external torch.hub repository; pass "
                               "backbone=<object with infer(frame | None, use_amp) / reset_state / metric_depth>")
        if not hasattr(net, "metric_depth"):
            net.metric_depth = self.metric_depth
        bound = resolution or 392
        net.prep_lower_bound = bound + (-bound) % 14             # e.g. the GUI's 512 -> 518
        self._ledger = _OnlineLedger(net)
        return net

    def _book(self):
        if self._ledger is None or self._ledger.net is not self.model:
            self._ledger = _OnlineLedger(self.model)
        return self._ledger

    def reset_state(self):
        self._book().restart()

    def reset(self):
        self.reset_state()
        self.reset_ema()

    def _prepare(self, x):
        return batch_preprocess(x.to(self.device), self.model.prep_lower_bound, metric_depth=self.model.metric_depth,
                                limit_resolution=self.limit_resolution)

    def _finish(self, maps, edge_dilation, depth_aa, enable_amp):
        """finished raw maps of the network -> post-processed [n,1,h,w]"""
        return postprocess(torch.stack(maps), edge_dilation=edge_dilation, depth_aa=depth_aa, metric_depth=self.model.metric_depth,
                           force_disparity=self.force_disparity, enable_amp=enable_amp)

    def _normalised(self, maps, edge_dilation, depth_aa, enable_amp):
        if not maps:
            return []
        done = (self.minmax_normalize_chw(d) for d in self._finish(maps, edge_dilation, depth_aa, enable_amp))
        return [d for d in done if d is not None]

    def _aa(self, flag):
        """``flush_with_normalize`` (reference :223-225): a bool selects the model's own DepthAA, anything else passes through."""
        return flag if not isinstance(flag, bool) and flag is not None else (self.depth_aa if flag else None)

    @torch.inference_mode()
    def infer(self, x, enable_amp=True, edge_dilation=0, **kwargs):
        """A single image through the video network (reference :166-190, marked "DONT USE THIS" there): one frame in, drain."""
        if not torch.is_tensor(x):
            raise ValueError("infer expects a CHW or BCHW float tensor in [0,1]")
        single = x.ndim == 3
        self.reset()
        book = self._book()
        book.feed(self._prepare(x.unsqueeze(0) if single else x)[0], enable_amp)
        book.n_out = 0                      # the reference ignores what this first call returns (:181-183): the drain owes one frame
        out = self._finish(book.drain(enable_amp), edge_dilation, None, enable_amp)
        self.reset()
        return out.squeeze(0) if single else out

    @torch.inference_mode()
    def infer_with_normalize(self, x, pts, reset_pts, enable_amp=True, edge_dilation=0, depth_aa=None, **kwargs):
        assert x.ndim == 4
        aa = self.depth_aa if depth_aa else None          # reference :195: ANY truthy value selects the model's own DepthAA here
        book, out = self._book(), []
        for frame, t in zip(self._prepare(x), pts):
            out += self._normalised(book.feed(frame, enable_amp), edge_dilation, aa, enable_amp)
            if t in reset_pts:                     # scene cut: nothing of this scene may leak into the next one
                out += self.flush_with_normalize(enable_amp=enable_amp, edge_dilation=edge_dilation, depth_aa=aa)
                self.reset()
        return out

    @torch.inference_mode()
    def flush_with_normalize(self, enable_amp=True, edge_dilation=0, depth_aa=None):
        tail = self._book().drain(enable_amp)
        if not tail:
            return []
        return self._normalised(tail, edge_dilation, self._aa(depth_aa), enable_amp) + self.flush_minmax_normalize()

    @classmethod
    def get_name(cls):
        return "VideoDepthAnything"

    def is_image_supported(self):
        return False

    @classmethod
    def supported(cls, model_type):
        return model_type in NAME_MAP

    def is_metric(self):
        return self.metric_depth and not self.force_disparity

    @classmethod
    def multi_gpu_supported(cls, model_type):
        return False
