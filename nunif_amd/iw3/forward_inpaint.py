"""``--method forward_inpaint`` on the HIP engine: un-filled depth-ordered forward warp + hole masks, then an inpaint net fills
the holes of each eye.  Mirrors ``iw3/forward_inpaint.py`` (reference): ``forward_right`` / ``forward_left`` :18-40 (mask > 0,
``mask_closing``, ``dilate_outer`` / ``dilate_inner``, ``model.infer``; the left eye is processed mirrored),
``ForwardInpaintImage`` :43-103, ``ForwardInpaintVideo`` :106-232 (12-frame ``FrameQueue``, 3 frames of temporal context on
either side) and the ``ForwardInpaint`` mode switch :235-300.

Every step is a kernel of the engine — ``nunif_hip_forward_warp`` (warp + masks in one row-local pass),
``nunif_hip_mask_morphology``, ``nunif_hip_light_inpaint_infer`` — this file is only their order.  Models are passed in as
objects (the reference downloads them by name)."""
import torch

from . import _ops
from .dilation import dilate_inner, dilate_outer, mask_closing
from .forward_warp import apply_divergence_forward_warp
from .inpaint_utils import FrameQueue


def _hole_mask(mask, inner_dilation, outer_dilation, base_width):
    mask = mask_closing(mask > 0)
    mask = dilate_outer(mask, n_iter=outer_dilation, base_width=base_width)
    return dilate_inner(mask, n_iter=inner_dilation, base_width=base_width)


def forward_right(model, right_eye, right_mask, inner_dilation, outer_dilation, base_width):
    return model.infer(right_eye, _hole_mask(right_mask, inner_dilation, outer_dilation, base_width))


def forward_left(model, left_eye, left_mask, inner_dilation, outer_dilation, base_width):
    left_eye, left_mask = left_eye.flip(-1), left_mask.flip(-1)           # right-view base (:31-32)
    return model.infer(left_eye, _hole_mask(left_mask, inner_dilation, outer_dilation, base_width)).flip(-1)


def _limit_width(x, max_width):
    """:70-77 — cap the working width (even sizes), bilinear antialias."""
    if max_width is not None and x.shape[-1] > max_width:
        if max_width % 2 != 0:
            max_width += 1
        new_h = int((max_width / x.shape[-1]) * x.shape[-2])
        if new_h % 2 != 0:
            new_h += 1
        x = _ops.resize_aa(x, (new_h, max_width), mode="bilinear", align_corners=False)
    return x


def _warp(x, depth, divergence, convergence, synthetic_view):
    return apply_divergence_forward_warp(x, depth, divergence=divergence, convergence=convergence,
                                         synthetic_view=synthetic_view, return_mask=True, width_base=False)


class ForwardInpaintImage:
    def __init__(self, model):
        """model: ``LightInpaintV1`` on the HIP engine."""
        self.model = model

    def reset(self):
        pass

    def flush(self, enable_amp=True):
        return None, None

    def infer(self, x, depth, divergence, convergence, synthetic_view="both", inner_dilation=0, outer_dilation=0,
              max_width=None, **_kwargs):
        x = _limit_width(x, max_width)
        left_eye, right_eye, left_mask, right_mask = _warp(x, depth, divergence, convergence, synthetic_view)
        kw = dict(inner_dilation=inner_dilation, outer_dilation=outer_dilation, base_width=depth.shape[-1])
        if synthetic_view in ("both", "left"):
            left_eye = forward_left(self.model, left_eye, left_mask, **kw)
        if synthetic_view in ("both", "right"):
            right_eye = forward_right(self.model, right_eye, right_mask, **kw)
        return left_eye, right_eye

    forward = infer
    __call__ = infer


class ForwardInpaintVideo:
    """:106-232.  ``infer`` takes a batch of consecutive frames (1 or 3, so that the queue lands exactly on 12) and returns
    the frames that have full temporal context — ``(None, None)`` while the queue fills; ``flush`` pads with copies of the
    last frame and returns the rest."""

    def __init__(self, model, pre_padding=3, post_padding=3):
        self.model, self.model_seq = model, 12
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

    def infer(self, x, depth, divergence, convergence, synthetic_view="both", inner_dilation=0, outer_dilation=0,
              max_width=None, **_kwargs):
        assert x.shape[0] <= self.model_seq                     # the queue must not grow past one window
        x = _limit_width(x, max_width)
        self.synthetic_view = synthetic_view
        self.inner_dilation, self.outer_dilation = inner_dilation, outer_dilation
        self.base_width = depth.shape[-1]
        if self.frame_queue is None:
            self.frame_queue = FrameQueue(synthetic_view=synthetic_view, seq=self.model_seq, height=x.shape[-2],
                                          width=x.shape[-1], dtype=x.dtype, device=x.device)
        left_eye, right_eye, left_mask, right_mask = _warp(x, depth, divergence, convergence, synthetic_view)
        for i in range(left_eye.shape[0]):
            repeat = self.pre_padding + 1 if self.frame_queue.empty() else 1
            for _ in range(repeat):
                if synthetic_view == "both":
                    self.frame_queue.add(left_eye[i], right_eye[i], left_mask=left_mask[i], right_mask=right_mask[i])
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


class ForwardInpaint:
    """The side-model object ``iw3.utils.apply_divergence`` drives for ``--method forward_inpaint`` (:235-300):
    ``.infer`` / ``.flush`` / ``.reset`` / ``.set_mode("image" | "video")``.  ``video_model`` (a ``LightVideoInpaintV1``) is
    optional: without it only the image mode exists."""

    def __init__(self, model, video_model=None):
        self.model = [ForwardInpaintImage(model), ForwardInpaintVideo(video_model) if video_model is not None else None]
        self.mode = 0

    def set_mode(self, mode):
        assert mode in {"video", "image"}
        if mode == "video" and self.model[1] is None:
            raise NotImplementedError("no video inpaint model was given (ForwardInpaint(image_model, video_model))")
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
    def infer(self, x, depth, divergence, convergence, synthetic_view="both", inner_dilation=0, outer_dilation=0,
              max_width=None, enable_amp=True, **_kwargs):
        return self.model[self.mode].infer(x, depth, divergence=divergence, convergence=convergence,
                                           synthetic_view=synthetic_view, inner_dilation=inner_dilation,
                                           outer_dilation=outer_dilation, max_width=max_width, **_kwargs)

    @torch.inference_mode()
    def flush(self, enable_amp=True):
        return self.model[self.mode].flush()
