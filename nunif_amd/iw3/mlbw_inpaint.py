"""Image-mode MLBW + inpaint on the HIP engine.  Mirrors ``iw3/mlbw_inpaint.py``: ``apply_divergence`` :38-75 (mask-MLBW warp
of both eyes with ``return_mask=True``), ``forward_right`` / ``forward_left`` :21-35 (hole mask post-processing, inpaint; the
left eye is processed mirrored), ``MLBWInpaintImage`` :78-157 and the ``MLBWInpaint`` mode switch :296-360.  Models are
passed in as objects (the reference downloads them); the video mode (12-frame ``FrameQueue`` + ``LightVideoInpaintV1``) is not
on the engine yet."""
import torch

from . import _ops
from .backward_warp import apply_divergence_nn_delta_weight, postprocess_hole_mask

MASK_MLBW_THRESHOLD = 0.15


def forward_right(model, right_eye, right_mask, inner_dilation, outer_dilation, base_width):
    right_mask = postprocess_hole_mask(right_mask, target_size=right_eye.shape[-2:], threshold=MASK_MLBW_THRESHOLD,
                                       inner_dilation=inner_dilation, outer_dilation=outer_dilation)
    return model.infer(right_eye, right_mask)


def forward_left(model, left_eye, left_mask, inner_dilation, outer_dilation, base_width):
    left_eye, left_mask = left_eye.flip(-1), left_mask.flip(-1)
    left_mask = postprocess_hole_mask(left_mask, target_size=left_eye.shape[-2:], threshold=MASK_MLBW_THRESHOLD,
                                      inner_dilation=inner_dilation, outer_dilation=outer_dilation)
    return model.infer(left_eye, left_mask).flip(-1)


def apply_divergence(model, c, depth, divergence, convergence, preserve_screen_border, synthetic_view, enable_amp):
    kw = dict(convergence=convergence, steps=1, preserve_screen_border=preserve_screen_border, enable_amp=enable_amp,
              return_mask=True)
    if synthetic_view == "both":
        left_eye, left_mask = apply_divergence_nn_delta_weight(model, c, depth, divergence=divergence, shift=-1, **kw)
        right_eye, right_mask = apply_divergence_nn_delta_weight(model, c, depth, divergence=divergence, shift=1, **kw)
    elif synthetic_view == "right":
        left_eye, left_mask = c, None
        right_eye, right_mask = apply_divergence_nn_delta_weight(model, c, depth, divergence=divergence * 2, shift=1, **kw)
    else:
        left_eye, left_mask = apply_divergence_nn_delta_weight(model, c, depth, divergence=divergence * 2, shift=-1, **kw)
        right_eye, right_mask = c, None
    return left_eye, right_eye, left_mask, right_mask


class MLBWInpaintImage:
    def __init__(self, model, mask_mlbw):
        """model: ``LightInpaintV1`` (HIP), mask_mlbw: ``MLBW(hole_mask=True)`` (HIP) with ``delta_output = True``."""
        self.model, self.mask_mlbw = model, mask_mlbw
        self.mask_mlbw.delta_output = True

    def reset(self):
        pass

    def flush(self, enable_amp=True):
        return None, None

    def infer(self, x, depth, divergence, convergence, preserve_screen_border=False, synthetic_view="both",
              inner_dilation=0, outer_dilation=0, max_width=None, enable_amp=True, **_kwargs):
        if max_width is not None and x.shape[-1] > max_width:
            if max_width % 2 != 0:
                max_width += 1
            new_w = max_width
            new_h = int((max_width / x.shape[-1]) * x.shape[-2])
            if new_h % 2 != 0:
                new_h += 1
            x = _ops.resize_aa(x, (new_h, new_w), mode="bilinear", align_corners=False)
        left_eye, right_eye, left_mask, right_mask = apply_divergence(
            self.mask_mlbw, x, depth, divergence=divergence, convergence=convergence,
            preserve_screen_border=preserve_screen_border, synthetic_view=synthetic_view, enable_amp=enable_amp)
        kw = dict(inner_dilation=inner_dilation, outer_dilation=outer_dilation, base_width=depth.shape[-1])
        if synthetic_view in ("both", "left"):
            left_eye = forward_left(self.model, left_eye, left_mask, **kw)
        if synthetic_view in ("both", "right"):
            right_eye = forward_right(self.model, right_eye, right_mask, **kw)
        return left_eye, right_eye

    forward = infer
    __call__ = infer


class MLBWInpaint:
    """The side-model object ``iw3.utils`` drives (``.infer`` / ``.flush`` / ``.reset`` / ``.set_mode``): image mode only."""

    def __init__(self, model, mask_mlbw):
        self.image = MLBWInpaintImage(model, mask_mlbw)
        self.mode = "image"

    def set_mode(self, mode):
        assert mode in {"video", "image"}
        if mode == "video":
            raise NotImplementedError("the video inpaint mode (FrameQueue + LightVideoInpaintV1) is not on the HIP engine yet")
        self.mode = mode

    def reset(self):
        self.image.reset()

    @torch.inference_mode()
    def infer(self, *args, **kwargs):
        return self.image.infer(*args, **kwargs)

    def flush(self, enable_amp=True):
        return self.image.flush(enable_amp=enable_amp)
