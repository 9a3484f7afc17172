"""Oracle: iw3 per-frame glue (torch CPU fp32) — mappers, SBS compose + quantise, frame <-> tensor, EMA scaler.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Follows ``iw3/mapper.py`` :7-61 (pointwise maps), ``iw3/utils.py`` ``postprocess_image`` :430-487 (the default
SBS / TB / cross-eyed / half-SBS branches; ``TF.resize(BICUBIC, antialias=True)`` == ``F.interpolate(bicubic,
align_corners=False, antialias=True)``), ``nunif/utils/video.py`` ``to_tensor`` :218-223 / ``from_tensor`` :236-245
and ``iw3/depth_scaler.py`` ``EMAMinMaxScaler`` :64-142.
"""
import math

import torch
import torch.nn.functional as F


def softplus01_legacy(x, c=6):
    lo = math.log(1 + math.exp(0 * 12.0 - c)) / (12 - c)
    hi = math.log(1 + math.exp(1 * 12.0 - c)) / (12 - c)
    return (torch.log(1. + torch.exp(x * 12.0 - c)) / (12 - c) - lo) / (hi - lo)


def softplus01(x, bias, scale):
    lo = math.log(1 + math.exp((0 - bias) * scale))
    hi = math.log(1 + math.exp((1 - bias) * scale))
    return (torch.log(1. + torch.exp((x - bias) * scale)) - lo) / (hi - lo)


def inv_softplus01(x, bias, scale):
    f = lambda t: ((t - bias) * scale).expm1().clamp(min=1e-6).log()    # noqa: E731
    lo, hi = f(torch.zeros(1)), f(torch.ones(1))
    return (f(x) - lo) / (hi - lo)


def distance_to_disparity(x, c):
    c1 = 1.0 + c
    lo = c / c1
    return ((c / (c1 - x)) - lo) / (1.0 - lo)


def shift_relative_depth(x, min_distance, max_distance=16):
    pmax = min_distance + max_distance
    a, b = 1.0 / pmax, (1.0 / min_distance) - (1.0 / pmax)
    dist = (1.0 - min_distance) + 1 / (a + b * x)
    lo = 1.0 / (max_distance + 1)
    return (1.0 / dist - lo) / (1.0 - lo)


MAPPERS = {
    "none": lambda x: x, "pow2": lambda x: x ** 2, "softplus": softplus01_legacy,
    "softplus2": lambda x: softplus01_legacy(x) ** 2,
    "mul_1": lambda x: softplus01(x, 0.343, 12), "mul_2": lambda x: softplus01(x, 0.515, 12),
    "mul_3": lambda x: softplus01(x, 0.687, 12),
    "inv_mul_1": lambda x: inv_softplus01(x, -0.002102, 7.8788), "inv_mul_2": lambda x: inv_softplus01(x, -0.0003, 6.2626),
    "inv_mul_3": lambda x: inv_softplus01(x, -0.0001, 3.4343),
    "shift_30": lambda x: shift_relative_depth(x, 3.0), "shift_14": lambda x: shift_relative_depth(x, 1.4),
    "shift_045": lambda x: shift_relative_depth(x, 0.45),
    "div_25": lambda x: distance_to_disparity(x, 2.5), "div_6": lambda x: distance_to_disparity(x, 0.6),
    "div_1": lambda x: distance_to_disparity(x, 0.1),
}


def compose(left, right, layout="sbs", half=False):
    if half and layout == "tb":
        size = (left.shape[1] // 2, left.shape[2])
    elif half:
        size = (left.shape[1], left.shape[2] // 2)
    if half:
        left, right = (F.interpolate(e[None], size=size, mode="bicubic", align_corners=False, antialias=True)[0]
                       for e in (left, right))
    if layout == "tb":
        out = torch.cat([left, right], 1)
    elif layout == "cross_eyed":
        out = torch.cat([right, left], 2)
    else:
        out = torch.cat([left, right], 2)
    return out.clamp(0., 1.)


def to_frame(x, bits=8):
    maxv = 255.0 if bits == 8 else 65535.0
    q = (x.permute(1, 2, 0).contiguous() * maxv).round()
    return q.to(torch.uint8) if bits == 8 else q.to(torch.int32)


def to_tensor(frame, bits=8):
    return frame.permute(2, 0, 1).contiguous().float() / (255 if bits == 8 else 65535)


class EMAScaler:
    """Restatement of EMAMinMaxScaler (minmax mode) over plain python floats."""

    def __init__(self, decay, buffer_size):
        self.decay, self.n = float(decay), int(buffer_size)
        self.reset()

    def reset(self):
        self.ring, self.count, self.queue, self.lo, self.hi = [0.0] * (2 * self.n), 0, [], None, None

    def _norm(self, f, lo, hi):
        return ((f - lo) / (hi - lo)).clamp(0, 1) if hi - lo > 0 else f.clamp(0, 1)

    def update(self, frame):
        lo, hi = frame.min().item(), frame.max().item()
        self.queue.append(frame)
        if self.count == 0:
            self.ring = [lo, hi] * self.n
            self.count = 2
        else:
            for v in (lo, hi):
                self.ring[self.count % (2 * self.n)] = v
                self.count += 1
        if self.count < 2 * self.n:
            return None
        wlo, whi = min(self.ring), max(self.ring)
        if self.lo is None:
            self.lo, self.hi = wlo, whi
        else:
            self.lo = self.decay * self.lo + (1 - self.decay) * wlo
            self.hi = self.decay * self.hi + (1 - self.decay) * whi
        return self._norm(self.queue.pop(0), self.lo, self.hi)

    def flush(self):
        if not self.queue:
            self.reset()
            return []
        lo, hi = (min(self.ring), max(self.ring)) if self.lo is None else (self.lo, self.hi)
        out = [self._norm(f, lo, hi) for f in self.queue]
        self.reset()
        return out
