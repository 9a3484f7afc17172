"""Depth normalisation.  Mirrors ``iw3/depth_scaler.py`` (reference): ``minmax_normalize`` :4-17, ``max_normalize``
:20-30, ``MinMaxBuffer`` :33-61 and ``EMAMinMaxScaler`` :64-142 (look-ahead ring of 2N min/max samples, EMA over the
window extrema, frame queue, flush on scene cut).  The per-frame reductions and the normalisation run on the HIP
engine (``nunif_hip_minmax_normalize``); the recurrence itself is sequential scalar state and stays on the host, as in
the reference (SURVEY.md §7 "Stateful normalisation")."""
import torch

from . import _ops


def minmax_normalize(frame, min_value=None, max_value=None):
    if min_value is None and max_value is None:        # the frame's own range (EMA off)
        single = frame.dim() == 3
        y = _ops.minmax_normalize(frame.unsqueeze(0) if single else frame)
        return y[0] if single else y
    if torch.is_tensor(min_value) and min_value.device != frame.device:
        # in-process multi-device (FrameCallbackPool over several GPUs): the scaler's running extrema live on the device of
        # the first frame it saw; a frame from another device gets its own copy of the two scalars
        min_value, max_value = min_value.to(frame.device), max_value.to(frame.device)
    scale = max_value - min_value
    if scale > 0:
        return ((frame - min_value) / scale).clamp(0.0, 1.0)
    return frame.clamp(0.0, 1.0)


def max_normalize(frame, min_value, max_value):
    if torch.is_tensor(max_value) and max_value.device != frame.device:
        max_value = max_value.to(frame.device)
    if max_value > 0:
        return (frame / max_value).clamp(0.0, 1.0)
    return frame.clamp(0.0, 1.0)


class MinMaxBuffer():
    def __init__(self, size, dtype, device):
        assert size > 0
        self.count = 0
        self.size = size * 2
        self.data = torch.zeros(self.size, dtype=dtype, device=device)

    def add(self, min_value, max_value):
        if self.count == 0:
            self.data[0::2] = min_value
            self.data[1::2] = max_value
            self.count = 2
        else:
            for v in (min_value, max_value):
                self.data[self.count % self.size] = v
                self.count += 1

    def is_filled(self):
        return self.count >= self.size

    def get_minmax(self):
        return self.data.amin(), self.data.amax()


class EMAMinMaxScaler():
    def __init__(self, decay=0, buffer_size=1, mode="minmax"):
        assert mode in {"minmax", "max"}
        self.normalize = {"minmax": minmax_normalize, "max": max_normalize}[mode]
        assert buffer_size > 0
        self.frame_queue = []
        self.reset(decay=decay, buffer_size=buffer_size)

    def reset(self, decay=None, buffer_size=None, **kwargs):
        if decay is not None:
            self.decay = float(decay)
        if buffer_size is not None:
            self.buffer_size = int(buffer_size)
        self.min_value = self.max_value = None
        self.frame_queue = []
        self.minmax_buffer = None

    def get_minmax(self):
        assert self.minmax_buffer is not None and self.minmax_buffer.is_filled()
        return self.minmax_buffer.get_minmax()

    def __call__(self, frame, return_minmax=False):
        return self.update(frame, return_minmax=return_minmax)

    def update(self, frame, return_minmax=False):
        if self.minmax_buffer is None:
            self.minmax_buffer = MinMaxBuffer(self.buffer_size, dtype=frame.dtype, device=frame.device)
        self.frame_queue.append(frame)
        lo_f, hi_f = frame.amin(), frame.amax()
        if lo_f.device != self.minmax_buffer.data.device:
            lo_f, hi_f = lo_f.to(self.minmax_buffer.data.device), hi_f.to(self.minmax_buffer.data.device)
        self.minmax_buffer.add(lo_f, hi_f)
        if not self.minmax_buffer.is_filled():
            return (None, None, None) if return_minmax else None
        lo, hi = self.get_minmax()
        if self.min_value is None:
            self.min_value, self.max_value = lo, hi
        else:
            self.min_value = self.decay * self.min_value + (1. - self.decay) * lo
            self.max_value = self.decay * self.max_value + (1. - self.decay) * hi
        out = self.normalize(self.frame_queue.pop(0), self.min_value, self.max_value)
        return (out, self.min_value, self.max_value) if return_minmax else out

    def flush(self, return_minmax=False):
        if not self.frame_queue:
            self.reset()
            return []
        if self.min_value is None:
            lo, hi = self.minmax_buffer.get_minmax()
        else:
            lo, hi = self.min_value, self.max_value
        frames = [self.normalize(f, lo, hi) for f in self.frame_queue]
        if return_minmax:
            frames = [(f, lo, hi) for f in frames]
        self.reset()
        return frames
