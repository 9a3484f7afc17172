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


class MinMaxBuffer:
    """The look-ahead ring: the (min, max) of the last ``size`` frames as ``size`` PAIRS (the reference keeps the same 2 * size
    scalars interleaved in one flat vector, :33-61).  The very first frame fills every pair, so the ring is "filled" after
    ``size`` frames."""

    def __init__(self, size, dtype, device):
        assert size > 0
        self.pairs = torch.zeros((size, 2), dtype=dtype, device=device)
        self.seen = 0

    @property
    def data(self):                      # the reference's flat view (min0, max0, min1, max1, ...)
        return self.pairs.view(-1)

    @property
    def count(self):                     # the reference counts scalars, starting at 2 after the first frame
        return 2 * self.seen

    @property
    def size(self):
        return self.pairs.numel()

    def add(self, min_value, max_value):
        pair = torch.stack([min_value.reshape(()), max_value.reshape(())]).to(self.pairs)
        if self.seen == 0:
            self.pairs[:] = pair
        else:
            self.pairs[self.seen % self.pairs.shape[0]] = pair
        self.seen += 1

    def is_filled(self):
        return self.seen >= self.pairs.shape[0]

    def get_minmax(self):
        return self.pairs.amin(), self.pairs.amax()


class EMAMinMaxScaler():
    """Reference :64-142.  Host tensors run the reference's arithmetic in torch (bit-identical to the live reference class in the
    CPU tests); DEVICE tensors run the same arithmetic as four HIP kernels on a small device state block
    (``nunif_hip_minmax`` / ``ema_scaler_push`` / ``ema_scaler_ring_minmax`` / ``range_normalize``): no ATen reduce / fill /
    elementwise kernels and no ``if scale > 0`` host synchronisation per frame.  The data-independent bookkeeping (ring
    count, "filled", "an EMA value exists") is host state in both paths."""

    def __init__(self, decay=0, buffer_size=1, mode="minmax"):
        assert mode in {"minmax", "max"}
        self.mode = mode
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
        # device path: [ring 2N | min_value | max_value] + the host-side mirror of MinMaxBuffer.count
        self._dev_state = None
        self._dev_count = 0
        self._dev_has_value = False

    def get_minmax(self):
        assert self.minmax_buffer is not None and self.minmax_buffer.is_filled()
        return self.minmax_buffer.get_minmax()

    def __call__(self, frame, return_minmax=False):
        return self.update(frame, return_minmax=return_minmax)

    # -- device path ----------------------------------------------------------------------------------------------------
    def _on_device(self, frame):
        return frame.is_cuda and frame.dtype == torch.float32 and self.minmax_buffer is None

    def _dev_lohi(self, device=None):
        """(min_value, max_value) as a 2-element device tensor; a frame that lives on ANOTHER device than the state (the
        in-process multi-device pool deals batches round-robin) gets its own copy of the two scalars."""
        n2 = 2 * self.buffer_size
        lohi = self._dev_state[n2:n2 + 2]
        return lohi if device is None or device == lohi.device else lohi.to(device)

    def _dev_update(self, frame, return_minmax):
        n2 = 2 * self.buffer_size
        if self._dev_state is None:
            self._dev_state = torch.empty(n2 + 2, dtype=torch.float32, device=frame.device)
        self.frame_queue.append(frame)
        keys = _ops.minmax_keys(frame.reshape(1, -1))
        if keys.device != self._dev_state.device:
            keys = keys.to(self._dev_state.device)
        filled = (2 if self._dev_count == 0 else self._dev_count + 2) >= n2
        _ops.ema_scaler_push(self._dev_state, keys, n2, self._dev_count, filled, not self._dev_has_value, self.decay)
        self._dev_count = 2 if self._dev_count == 0 else self._dev_count + 2
        if not filled:
            return (None, None, None) if return_minmax else None
        self._dev_has_value = True
        first = self.frame_queue.pop(0)
        out = _ops.range_normalize(first, self._dev_lohi(first.device), max_mode=self.mode == "max")
        if return_minmax:
            lohi = self._dev_lohi().clone()
            return out, lohi[0], lohi[1]
        return out

    def _dev_flush(self, return_minmax):
        if not self._dev_has_value:
            _ops.ema_scaler_ring_minmax(self._dev_state, 2 * self.buffer_size)
        lohi = self._dev_lohi()
        frames = [_ops.range_normalize(f, self._dev_lohi(f.device), max_mode=self.mode == "max") for f in self.frame_queue]
        if return_minmax:
            c = lohi.clone()
            frames = [(f, c[0], c[1]) for f in frames]
        self.reset()
        return frames

    # -- host path (host tensors, or a stream that did not start on the device path) ---------------------------------------
    # Same arithmetic as the reference in the same order (decay * old + (1 - decay) * new on 0-d tensors, then
    # (frame - lo) / (hi - lo) and a clamp), so the CPU tests can demand bit-identical output from the live reference class.
    def _blend(self, old, new):
        return new if old is None else self.decay * old + (1. - self.decay) * new

    def _answer(self, frames, lo, hi, return_minmax):
        done = [self.normalize(f, lo, hi) for f in frames]
        return [(f, lo, hi) for f in done] if return_minmax else done

    def update(self, frame, return_minmax=False):
        if self._on_device(frame):
            return self._dev_update(frame, return_minmax)
        if self._dev_state is not None:
            raise RuntimeError("EMAMinMaxScaler: a stream started with device frames cannot continue with host frames; reset()")
        ring = self.minmax_buffer
        if ring is None:
            ring = self.minmax_buffer = MinMaxBuffer(self.buffer_size, dtype=frame.dtype, device=frame.device)
        self.frame_queue.append(frame)
        ring.add(frame.amin().to(ring.pairs.device), frame.amax().to(ring.pairs.device))
        if not ring.is_filled():                            # still looking ahead: nothing leaves yet
            return (None, None, None) if return_minmax else None
        lo, hi = ring.get_minmax()
        self.min_value, self.max_value = self._blend(self.min_value, lo), self._blend(self.max_value, hi)
        out = self._answer([self.frame_queue.pop(0)], self.min_value, self.max_value, return_minmax)[0]
        return out

    def flush(self, return_minmax=False):
        if not self.frame_queue:
            self.reset()
            return []
        if self._dev_state is not None:
            return self._dev_flush(return_minmax)
        # a scene shorter than the look-ahead never produced an EMA value: the ring's own extrema stand in
        lo, hi = (self.min_value, self.max_value) if self.min_value is not None else self.minmax_buffer.get_minmax()
        frames = self._answer(self.frame_queue, lo, hi, return_minmax)
        self.reset()
        return frames
