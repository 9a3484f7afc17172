"""Image-mode MLBW + inpaint on the HIP engine.  Mirrors ``iw3/mlbw_inpaint.py``: ``apply_divergence`` :38-75 (mask-MLBW warp
of both eyes with ``return_mask=True``), ``forward_right`` / ``forward_left`` :21-35 (hole mask post-processing, inpaint; the
left eye is processed mirrored), ``MLBWInpaintImage`` :78-157 and the ``MLBWInpaint`` mode switch :296-360.  Models are
passed in as objects (the reference downloads them).  ``MLBWInpaintVideo`` :160-293 keeps the last 12 warped frames in a
``FrameQueue`` (3 frames of temporal context on either side) and runs ``LightVideoInpaintV1`` whenever the queue is full."""
import torch

from . import _ops
from .backward_warp import apply_divergence_nn_delta_weight, postprocess_hole_mask
from .inpaint_utils import FrameQueue

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


class MLBWInpaintVideo:
    """Reference :160-293.  ``infer`` takes a batch of consecutive frames (sizes that let the queue land exactly on 12:
    1 or 3), returns the frames that have full temporal context — or ``(None, None)`` while the queue fills; ``flush`` pads the
    queue with copies of the last frame and returns the rest."""

    def __init__(self, model, mask_mlbw, pre_padding=3, post_padding=3):
        self.model, self.mask_mlbw = model, mask_mlbw
        self.mask_mlbw.delta_output = True
        self.model_seq = 12
        self.pre_padding, self.post_padding = pre_padding, post_padding
        self.frame_queue = None
        self.synthetic_view = self.inner_dilation = self.outer_dilation = self.base_width = None

    def reset(self):
        self.frame_queue = None

    def forward(self, flush=False):
        if not self.frame_queue.full():
            return None, None
        kw = dict(inner_dilation=self.inner_dilation, outer_dilation=self.outer_dilation, base_width=self.base_width)
        if self.synthetic_view == "both":
            left_eye, right_eye, left_mask, right_mask = self.frame_queue.get()
            left_eye = forward_left(self.model, left_eye, left_mask, **kw)
            right_eye = forward_right(self.model, right_eye, right_mask, **kw)
        elif self.synthetic_view == "right":
            left_eye, right_eye, right_mask = self.frame_queue.get()
            right_eye = forward_right(self.model, right_eye, right_mask, **kw)
            left_eye = left_eye.clone()
        else:
            left_eye, right_eye, left_mask = self.frame_queue.get()
            left_eye = forward_left(self.model, left_eye, left_mask, **kw)
            right_eye = right_eye.clone()
        if flush:
            left_eye, right_eye = left_eye[self.pre_padding:], right_eye[self.pre_padding:]
            self.frame_queue.clear()
        else:
            if self.post_padding > 0:
                left_eye = left_eye[self.pre_padding:-self.post_padding]
                right_eye = right_eye[self.pre_padding:-self.post_padding]
            elif self.pre_padding > 0:
                left_eye, right_eye = left_eye[self.pre_padding:], right_eye[self.pre_padding:]
            self.frame_queue.remove(self.model_seq - (self.pre_padding + self.post_padding))
        return left_eye, right_eye

    def infer(self, x, depth, divergence, convergence, preserve_screen_border=False, synthetic_view="both",
              inner_dilation=0, outer_dilation=0, max_width=None, enable_amp=True, **_kwargs):
        assert x.shape[0] <= self.model_seq
        if max_width is not None and x.shape[-1] > max_width:
            if max_width % 2 != 0:
                max_width += 1
            new_w = max_width
            new_h = int((max_width / x.shape[-1]) * x.shape[-2])
            if new_h % 2 != 0:
                new_h += 1
            x = _ops.resize_aa(x, (new_h, new_w), mode="bilinear", align_corners=False)
        self.synthetic_view = synthetic_view
        self.inner_dilation, self.outer_dilation = inner_dilation, outer_dilation
        self.base_width = depth.shape[-1]
        if self.frame_queue is None:
            self.frame_queue = FrameQueue(synthetic_view=synthetic_view, seq=self.model_seq, height=x.shape[-2],
                                          width=x.shape[-1], mask_height=depth.shape[-2], mask_width=depth.shape[-1],
                                          dtype=x.dtype, device=x.device)
        left_eye, right_eye, left_mask, right_mask = apply_divergence(
            self.mask_mlbw, x, depth, divergence=divergence, convergence=convergence,
            preserve_screen_border=preserve_screen_border, synthetic_view=synthetic_view, enable_amp=enable_amp)
        for i in range(left_eye.shape[0]):
            repeat = self.pre_padding + 1 if self.frame_queue.empty() else 1
            for _ in range(repeat):
                if synthetic_view == "both":
                    self.frame_queue.add(left_eye[i], right_eye[i], left_mask[i], right_mask[i])
                elif synthetic_view == "right":
                    self.frame_queue.add(left_eye[i], right_eye[i], right_mask=right_mask[i])
                else:
                    self.frame_queue.add(left_eye[i], right_eye[i], left_mask=left_mask[i])
        return self.forward()

    def flush(self, enable_amp=True):
        if self.frame_queue is None or self.frame_queue.empty():
            return None, None
        pad = self.frame_queue.fill()
        left_eye, right_eye = self.forward(flush=True)
        return (left_eye[:-pad], right_eye[:-pad]) if pad > 0 else (left_eye, right_eye)


class MLBWInpaint:
    """The side-model object ``iw3.utils`` drives (``.infer`` / ``.flush`` / ``.reset`` / ``.set_mode``), reference :296-360.
    ``video_model`` (a ``LightVideoInpaintV1``) is optional: without it only the image mode exists."""

    def __init__(self, model, mask_mlbw, video_model=None):
        self.model = [MLBWInpaintImage(model, mask_mlbw),
                      MLBWInpaintVideo(video_model, mask_mlbw) if video_model is not None else None]
        self.mode = 0

    def set_mode(self, mode):
        assert mode in {"video", "image"}
        if mode == "video" and self.model[1] is None:
            raise NotImplementedError("no video inpaint model was given (MLBWInpaint(image_model, mask_mlbw, video_model))")
        self.mode = 1 if mode == "video" else 0

    def reset(self):
        self.model[self.mode].reset()

    # torch.compile plumbing of the reference (compile / clear_compiled_model / compile_context, CompileContext in
    # iw3/inpaint_utils.py:191-203): the engine's nets are already native, so these cost nothing and change nothing
    def compile(self):
        pass

    def clear_compiled_model(self):
        pass

    def compile_context(self, enabled=True):
        import contextlib
        return contextlib.nullcontext()

    def train(self, mode=True):
        return self                      # inference only (the reference pins eval() the same way)

    def eval(self):
        return self

    @torch.inference_mode()
    def infer(self, *args, **kwargs):
        return self.model[self.mode].infer(*args, **kwargs)

    @torch.inference_mode()
    def flush(self, enable_amp=True):
        return self.model[self.mode].flush(enable_amp=enable_amp)
