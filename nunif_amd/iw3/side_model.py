"""One driver for every "warp both eyes, then let a net repair the holes" side model of iw3.

The reference has four near-identical classes for this — ``ForwardInpaintImage`` / ``ForwardInpaintVideo``
(``iw3/forward_inpaint.py:43-232``) and ``MLBWInpaintImage`` / ``MLBWInpaintVideo`` (``iw3/mlbw_inpaint.py:78-293``, which
carries a ``# TODO: Refactor with ForwardInpaintVideo`` at :155) — plus a ``FrameQueue`` of named buffers
(``iw3/inpaint_utils.py:98-188``).  What actually differs between the two families is two functions:

* ``warp(x, depth, divergence, convergence, synthetic_view, ...) -> (left, right, left_mask, right_mask)``
* ``hole_mask(raw_mask, eye_size, params) -> bool mask`` (how the warp's raw mask becomes the inpaint mask)

so here those two are parameters (:class:`SideModelSpec`) and everything else exists once:

* :class:`StereoWindow` — the temporal window as ONE object: per-eye frame and mask tracks of fixed capacity in HBM, a fill
  level, ``push`` / ``pad_with_last`` / ``slide``.
* :class:`ImageDriver` — warp, repair each synthesised eye, done.
* :class:`TemporalDriver` — warp, push frame by frame; WHENEVER the window is full the video net runs on it and the frames
  with ``lead`` past / ``lag`` future frames of context are emitted, then the window slides by ``capacity - lead - lag``.
  Because the check sits inside the push loop, any batch size works (the reference asserts ``B <= 12`` and relies on batch
  sizes that land exactly on a full queue; for those — 1 and 3 — the schedule here is the same one: the fixtures
  ``light_video_inpaint.npz`` / ``forward_inpaint.npz`` hold the reference's outputs call by call).
* :class:`SideModel` — the object ``iw3.utils.apply_divergence`` drives: ``infer`` / ``flush`` / ``reset`` /
  ``set_mode("image" | "video")`` and the ``torch.compile`` plumbing of the reference as no-ops (the nets are native).

Nothing is computed here; every tensor operation is a kernel of the engine reached through the two spec functions and the
inpaint nets' ``infer``.
"""
import contextlib
import os
from dataclasses import dataclass
from typing import Callable

import torch

from . import _ops


@dataclass
class HoleParams:
    """What a ``hole_mask`` function needs beside the raw mask (the reference threads these through as loose attributes)."""
    inner_dilation: int = 0
    outer_dilation: int = 0
    base_width: int = 0


@dataclass
class SideModelSpec:
    warp: Callable           # (x, depth, divergence, convergence, synthetic_view, **warp_kwargs) -> left, right, lmask, rmask
    hole_mask: Callable      # (raw_mask, (H, W) of the eye, HoleParams) -> mask the inpaint net takes
    mask_at_depth_size: bool = False     # True: the warp's masks live at the DEPTH map's resolution (MLBW hole logits)


def limit_width(x, max_width):
    """Cap the working width (``--inpaint-max-width``): even target sizes, antialiased bilinear
    (``iw3/mlbw_inpaint.py:121-128``)."""
    if max_width is None or x.shape[-1] <= max_width:
        return x
    new_w = max_width + (max_width & 1)
    new_h = int((new_w / x.shape[-1]) * x.shape[-2])
    new_h += new_h & 1
    return _ops.resize_aa(x, (new_h, new_w), mode="bilinear", align_corners=False)


def _repair(net, eye, raw_mask, spec, params, mirrored):
    """One eye through the inpaint net.  The nets are trained on the RIGHT view, so the left eye goes through mirrored
    (``forward_left``, ``iw3/forward_inpaint.py:29-40``)."""
    if mirrored and getattr(net, "supports_mirror_x", False) and os.environ.get("NUNIF_INPAINT_MIRROR", "1") != "0":
        # the engine reads and writes the picture mirrored (two flip passes over the eye saved: 0.8 GB each way per 12-frame 4K
        # window); only the raw mask — small at depth resolution — is flipped, its post-processing is directional
        return net.infer(eye, spec.hole_mask(raw_mask.flip(-1), eye.shape[-2:], params), mirror_x=True)
    if mirrored:
        eye, raw_mask = eye.flip(-1), raw_mask.flip(-1)
    out = net.infer(eye, spec.hole_mask(raw_mask, eye.shape[-2:], params))
    return out.flip(-1) if mirrored else out


def _repair_pair(net, left, right, lmask, rmask, view, spec, params):
    if view in ("both", "left"):
        left = _repair(net, left, lmask, spec, params, mirrored=True)
    if view in ("both", "right"):
        right = _repair(net, right, rmask, spec, params, mirrored=False)
    return left, right


class StereoWindow:
    """The last ``capacity`` stereo frames (and the masks of the synthesised eyes) in HBM, oldest first.

    The window is a ``capacity``-frame VIEW that travels along a longer buffer (``span`` frames, default 3 x capacity): ``slide``
    moves the view's origin instead of the frames, and only when the view reaches the end of the buffer are the kept frames
    copied back to the front — once every ``(span - capacity) / n`` slides, into a region the source cannot overlap (no staging
    clone).  Round 4 moved every kept frame on every slide, through a clone: 546 ``copyBuffer`` launches of 58.6 MB in the
    config-5 profile (``profiles/r04f_pmc_WRITE_SIZE.txt``).  ``get`` hands out a contiguous view either way."""

    def __init__(self, view, capacity, frame_shape, mask_shape, dtype, device, span=None):
        if span is None:
            span = int(os.environ.get("NUNIF_STEREO_WINDOW_SPAN", "3")) * capacity
        assert span >= 2 * capacity or span == capacity, "a copy-back must not overlap its source"
        self.view, self.capacity, self.span, self.level, self.start = view, capacity, span, 0, 0
        self.tracks = {"left": torch.zeros((span, *frame_shape), dtype=dtype, device=device),
                       "right": torch.zeros((span, *frame_shape), dtype=dtype, device=device)}
        if view in ("both", "left"):
            self.tracks["left_mask"] = torch.zeros((span, *mask_shape), dtype=dtype, device=device)
        if view in ("both", "right"):
            self.tracks["right_mask"] = torch.zeros((span, *mask_shape), dtype=dtype, device=device)

    def is_full(self):
        return self.level == self.capacity

    def is_empty(self):
        return self.level == 0

    def push(self, **frames):
        """One frame per track (tracks the window does not keep are ignored)."""
        if self.is_full():
            raise IndexError("StereoWindow.push on a full window")
        for name, buf in self.tracks.items():
            buf[self.start + self.level] = frames[name]
        self.level += 1

    def pad_with_last(self):
        """Repeat the newest frame until the window is full; returns the number of copies."""
        n = self.capacity - self.level
        if n > 0:
            a = self.start + self.level
            for buf in self.tracks.values():
                buf[a:self.start + self.capacity] = buf[a - 1]
            self.level = self.capacity
        return n

    def slide(self, n):
        """Drop the ``n`` oldest frames."""
        keep = self.level - n
        if keep < 0:
            raise IndexError(f"StereoWindow.slide({n}) with {self.level} frames")
        self.start += n
        self.level = keep
        if self.start + self.capacity > self.span:          # no room for a whole window behind the origin: back to the front
            if keep > 0:
                if self.start >= keep:
                    for buf in self.tracks.values():
                        buf[:keep] = buf[self.start:self.start + keep]
                else:                                        # (span == capacity: the round-4 behaviour)
                    for buf in self.tracks.values():
                        buf[:keep] = buf[self.start:self.start + keep].clone()
            self.start = 0

    def reset(self):
        self.level = self.start = 0

    def get(self, name):
        buf = self.tracks.get(name)
        return None if buf is None else buf[self.start:self.start + self.capacity]


class ImageDriver:
    def __init__(self, net, spec):
        self.net, self.spec = net, spec

    def reset(self):
        pass

    def flush(self, enable_amp=True):
        return None, None

    def infer(self, x, depth, divergence, convergence, synthetic_view="both", inner_dilation=0, outer_dilation=0,
              max_width=None, **warp_kwargs):
        x = limit_width(x, max_width)
        left, right, lmask, rmask = self.spec.warp(x, depth, divergence, convergence, synthetic_view, **warp_kwargs)
        params = HoleParams(inner_dilation, outer_dilation, depth.shape[-1])
        return _repair_pair(self.net, left, right, lmask, rmask, synthetic_view, self.spec, params)

    forward = infer
    __call__ = infer


class TemporalDriver:
    """``capacity`` = the video net's sequence length (12); a frame is emitted once it has ``lead`` frames before and ``lag``
    frames after it inside one window (the first frame of a stream is repeated ``lead`` times in front of itself, the last
    one padded behind itself at ``flush``)."""

    def __init__(self, net, spec, capacity=12, lead=3, lag=3):
        self.net, self.spec = net, spec
        self.capacity, self.lead, self.lag = capacity, lead, lag
        self.window = None
        self.view, self.params = None, HoleParams()

    def reset(self):
        self.window = None

    def _run_window(self):
        w = self.window
        left, right = w.get("left"), w.get("right")
        left, right = _repair_pair(self.net, left, right, w.get("left_mask"), w.get("right_mask"), self.view, self.spec, self.params)
        # an eye that is not synthesised is the window's own buffer: hand out a copy, the window keeps sliding underneath
        if self.view == "right":
            left = left.clone()
        elif self.view == "left":
            right = right.clone()
        return left, right

    def infer(self, x, depth, divergence, convergence, synthetic_view="both", inner_dilation=0, outer_dilation=0,
              max_width=None, **warp_kwargs):
        x = limit_width(x, max_width)
        self.view = synthetic_view
        self.params = HoleParams(inner_dilation, outer_dilation, depth.shape[-1])
        if self.window is None:
            mask_hw = depth.shape[-2:] if self.spec.mask_at_depth_size else x.shape[-2:]
            self.window = StereoWindow(synthetic_view, self.capacity, (3, *x.shape[-2:]), (1, *mask_hw), x.dtype, x.device)
        left, right, lmask, rmask = self.spec.warp(x, depth, divergence, convergence, synthetic_view, **warp_kwargs)
        emitted = []
        step = self.capacity - self.lead - self.lag
        for i in range(left.shape[0]):
            frame = dict(left=left[i], right=right[i], left_mask=None if lmask is None else lmask[i],
                         right_mask=None if rmask is None else rmask[i])
            for _ in range(self.lead + 1 if self.window.is_empty() else 1):
                self.window.push(**frame)
            if self.window.is_full():
                le, ri = self._run_window()
                emitted.append((le[self.lead:self.capacity - self.lag], ri[self.lead:self.capacity - self.lag]))
                self.window.slide(step)
        if not emitted:
            return None, None
        if len(emitted) == 1:
            return emitted[0]
        return torch.cat([e[0] for e in emitted]), torch.cat([e[1] for e in emitted])

    def flush(self, enable_amp=True):
        if self.window is None or self.window.is_empty():
            return None, None
        pad = self.window.pad_with_last()
        left, right = self._run_window()
        self.window.reset()
        end = self.capacity - pad
        return left[self.lead:end], right[self.lead:end]

    forward = infer
    __call__ = infer


class SideModel:
    """``.infer`` / ``.flush`` / ``.reset`` / ``.set_mode`` (``iw3/mlbw_inpaint.py:296-360``, ``iw3/forward_inpaint.py:235-300``)."""

    def __init__(self, spec, image_net, video_net=None, what="side model"):
        self.spec, self.what = spec, what
        self.drivers = {"image": ImageDriver(image_net, spec),
                        "video": TemporalDriver(video_net, spec) if video_net is not None else None}
        self.mode = "image"

    def set_mode(self, mode):
        assert mode in {"video", "image"}
        if self.drivers[mode] is None:
            raise NotImplementedError(f"{self.what}: no video inpaint model was given")
        self.mode = mode

    def reset(self):
        self.drivers[self.mode].reset()

    # torch.compile plumbing of the reference (``CompileContext``, iw3/inpaint_utils.py:191-203): native nets, nothing to do
    def compile(self):
        pass

    def clear_compiled_model(self):
        pass

    def compile_context(self, enabled=True):
        return contextlib.nullcontext()

    def train(self, mode=True):
        return self                      # inference only (the reference pins eval() the same way)

    def eval(self):
        return self

    @torch.inference_mode()
    def infer(self, x, depth, divergence, convergence, **kwargs):
        return self.drivers[self.mode].infer(x, depth, divergence, convergence, **self._infer_kwargs(kwargs))

    def _infer_kwargs(self, kwargs):
        return kwargs

    @torch.inference_mode()
    def flush(self, enable_amp=True):
        return self.drivers[self.mode].flush(enable_amp=enable_amp)
