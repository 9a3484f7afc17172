"""The Video-Depth-Anything network (streaming) on the HIP engine: the object the reference gets from
``torch.hub.load("nagadomi/Video-Depth-Anything_iw3:main", "VideoDepthAnythingStreaming", encoder=..., metric_depth=...)``
(``iw3/video_depth_anything_streaming_model.py:58-65``) and drives with ``model.infer_video_depth_one(frame, use_amp)`` (:94) and
``model.reset_state()`` (:75).

DINOv2 encoder + the DPT head with four temporal ("motion") modules; engine: ``nunif_amd/csrc/depth_anything.hip`` +
``depth_temporal.hip`` (K0 / V0 ring caches of the 32-frame attention window live in the engine handle, so one instance = one
stream; shard by scene or file, never by frame — SURVEY.md §8e).  The hub repository and its checkpoints are not reachable offline:
the architecture and the checkpoint key layout (``pretrained.*``, ``head.*``, ``head.motion_modules.{0..3}.*``) are the PUBLISHED
ones, restated in ``oracle/video_depth_anything_net.py`` — **parity unpinned** (INTEGRATION.md), the streaming cache policy in
particular.  ``VideoDepthAnythingStreamingModel.load_model(backbone=HipVideoDepthAnythingStreaming(state_dict))`` plugs it in.
"""
import torch

from .. import _hip
from .depth_anything_v2 import HipDepthAnythingV2


class HipVideoDepthAnythingStreaming(HipDepthAnythingV2):
    def __init__(self, state_dict, device="cuda:0", metric_depth=False):
        # the engine reads the DPT head under the Depth-Anything prefix `depth_head.`; the published VDA checkpoints say `head.`
        renamed = {("depth_head." + k[len("head."):] if k.startswith("head.") else k): v for k, v in state_dict.items()}
        if not any(k.startswith("depth_head.motion_modules.") for k in renamed):
            raise ValueError("not a Video-Depth-Anything checkpoint: no head.motion_modules.* tensors")
        super().__init__(renamed, device)
        if not _hip.lib().nunif_hip_depth_anything_is_temporal(self.handle):
            raise RuntimeError("the engine did not pick up the temporal modules")
        self._vda_state_dict = state_dict
        # the metric variants (Metric-Video-Depth-Anything) are the same network with other weights: ReLU output = distance
        self.metric_depth = bool(metric_depth)
        self.prep_lower_bound = 392
        self.frame_id = 0

    def replica(self, device):
        """Another STREAM of the same network (its own window state) — e.g. one per scene segment or per GPU."""
        return type(self)(self._vda_state_dict, device, self.metric_depth)

    def reset_state(self):
        _hip.check(_hip.lib().nunif_hip_depth_anything_reset_state(self.handle))
        self.frame_id = 0

    @torch.inference_mode()
    def infer_video_depth_one(self, frame, use_amp=True):
        """frame: [3, h, w] ImageNet-normalised, h and w multiples of 14 (``batch_preprocess``) -> [1, h, w].  ``use_amp`` is accepted
        for the hub signature; the engine computes in fp16 with fp32 accumulation either way."""
        if frame.dim() != 3:
            raise ValueError(f"infer_video_depth_one takes one CHW frame, got {tuple(frame.shape)}")
        self.frame_id += 1
        return self(frame.unsqueeze(0))

    @torch.inference_mode()
    def infer_video_depth_batch(self, frames, use_amp=True):
        """frames: [B, 3, h, w], B CONSECUTIVE frames of the stream -> [B, h, w]: what B calls of ``infer_video_depth_one`` return, in one
        pass of the engine (encoder, convs and Linears over the B frames at once; only the temporal attention steps frame by frame).
        Not part of the hub object's interface — ``VideoDepthAnythingStreamingModel.infer`` takes it when the network offers it.
        A batch must NOT span a scene cut: ``reset_state()`` takes effect between calls only (the reference resets between frames,
        ``iw3/video_depth_anything_streaming_model.py:74-75``), so callers split their batches at cuts — the per-frame route the
        reference takes for this model (``bind_single_frame_callback``, ``iw3/utils.py:1115``) hands over one frame per call."""
        if frames.dim() != 4:
            raise ValueError(f"infer_video_depth_batch takes BCHW frames, got {tuple(frames.shape)}")
        self.frame_id += frames.shape[0]
        return self(frames)
