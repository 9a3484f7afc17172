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


# ---- output formats ---------------------------------------------------------------------------------------------------------
# iw3/anaglyph.py :4-110 (seven red-cyan methods), iw3/equirectangular.py :7-40 (VR180), iw3/utils.py apply_rgbd :74-88,
# postprocess_padding :394-427 and the full postprocess_image :430-487.
def _luma(x):
    return x[0:1] * 0.299 + x[1:2] * 0.587 + x[2:3] * 0.114


def _srgb_decode(x):
    return torch.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def _srgb_encode(x):
    return torch.where(x <= 0.0031308, x * 12.92, 1.055 * x ** (1.0 / 2.4) - 0.055)


_DUBOIS_L = ((0.437, 0.449, 0.164), (-0.062, -0.062, -0.024), (-0.048, -0.050, -0.017))
_DUBOIS_R = ((-0.011, -0.032, -0.007), (0.377, 0.761, 0.009), (-0.026, -0.093, 1.234))


def anaglyph(left, right, kind):
    if kind == "color":
        return torch.cat([left[0:1], right[1:3]], 0)
    if kind == "gray":
        ry = _luma(right)
        return torch.cat([_luma(left), ry, ry], 0).clamp(0, 1)
    if kind == "half-color":
        return torch.cat([_luma(left), right[1:3]], 0).clamp(0, 1)
    if kind == "wimmer":
        return torch.cat([left[1:2] * 0.7 + left[2:3] * 0.3, right[1:3]], 0).clamp(0, 1)
    if kind == "wimmer2":
        def gb(e):
            return (e[1:2] + 0.45 * (e[0:1] - e[1:2]).clamp(min=0), e[2:3] + 0.25 * (e[0:1] - e[2:3]).clamp(min=0))
        (gl, bl), (gr, br) = gb(left), gb(right)
        return torch.cat([(0.75 * gl + 0.25 * bl) ** (1.0 / 1.6), gr, br], 0).clamp(0, 1)
    if kind in ("dubois", "dubois2"):
        ll, rl = _srgb_decode(left), _srgb_decode(right)
        rows = []
        for lm, rm in zip(_DUBOIS_L, _DUBOIS_R):
            a = (ll * torch.tensor(lm).view(3, 1, 1)).sum(0, keepdim=True)
            b = (rl * torch.tensor(rm).view(3, 1, 1)).sum(0, keepdim=True)
            if kind == "dubois":
                a, b = a.clamp(0, 1), b.clamp(0, 1)
            rows.append(a + b)
        return _srgb_encode(torch.cat(rows, 0).clamp(0, 1)).clamp(0, 1)
    raise ValueError(kind)


def equirectangular(c):
    h, w = c.shape[1:]
    edge = max(h, w)
    size = edge + edge // 2
    pw, ph = (size - w) // 2, (size - h) // 2
    c = F.pad(c, (pw, pw, ph, ph))
    H, W = c.shape[1:]
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    az, el = xs * (math.pi * 0.5), ys * (math.pi * 0.5)
    k = edge / size
    grid = torch.stack([k * torch.tan(az), k * (torch.tan(el) / torch.cos(az))], 2)
    return F.grid_sample(c[None], grid[None], mode="bicubic", padding_mode="zeros", align_corners=True)[0].clamp(0, 1)


def rgbd(im, depth):
    right = F.interpolate(depth[None], im.shape[-2:], mode="bicubic", antialias=True)[0]
    return im, right.expand_as(im)


def padding(left, right, pad, pad_mode):
    l = t = r = b = 0
    H, W = left.shape[1:]
    if pad_mode in ("tblr", "tb", "lr"):
        if "tb" in pad_mode:
            t = b = round(H * pad) // 2
        if "lr" in pad_mode:
            l = r = round(W * pad) // 2
    elif pad_mode == "top":
        t = round(H * pad)
    elif abs(16 / 9 - W / H) > 1e-3:
        if W / H > 16 / 9:
            t = b = (round(W / (16 / 9)) - H) // 2
        else:
            l = r = (round(H * (16 / 9)) - W) // 2
    return F.pad(left, (l, r, t, b)), F.pad(right, (l, r, t, b))


def postprocess_image(left, right, ipd_offset=0, pad=None, pad_mode=None, vr180=False, half_sbs=False, half_tb=False,
                      half_rgbd=False, rgbd_out=False, tb=False, cross_eyed=False, anaglyph_kind=None,
                      max_output_height=None, max_output_width=None, keep_aspect_ratio=False):
    ipd = int(abs(ipd_offset) * 0.01 * max(left.shape[-2:]))
    ipd -= ipd % 2
    if ipd > 0 and not (rgbd_out or half_rgbd):
        o, i = (ipd * 2, ipd) if ipd_offset > 0 else (ipd, ipd * 2)
        left, right = F.pad(left, (o, i, 0, 0)), F.pad(right, (i, o, 0, 0))
    if pad is not None or pad_mode == "16:9":
        left, right = padding(left, right, pad, pad_mode)
    bic = lambda e, size: F.interpolate(e[None], size=size, mode="bicubic", align_corners=False, antialias=True)[0]  # noqa: E731
    if vr180:
        left, right = equirectangular(left), equirectangular(right)
    elif half_sbs or half_rgbd:
        left, right = (bic(e, (e.shape[1], e.shape[2] // 2)) for e in (left, right))
    elif half_tb:
        left, right = (bic(e, (e.shape[1] // 2, e.shape[2])) for e in (left, right))
    if anaglyph_kind is not None:
        out = anaglyph(left, right, anaglyph_kind)
    elif tb or half_tb:
        out = torch.cat([left, right], 1).clamp(0, 1)
    elif cross_eyed:
        out = torch.cat([right, left], 2).clamp(0, 1)
    else:
        out = torch.cat([left, right], 2).clamp(0, 1)
    h, w = out.shape[1:]
    nw, nh = w, h
    if max_output_height is not None and nh > max_output_height:
        if keep_aspect_ratio:
            nw = int(max_output_height / nh * nw)
        nh = max_output_height
    if max_output_width is not None and nw > max_output_width:
        if keep_aspect_ratio:
            nh = int(max_output_width / nw * nh)
        nw = max_output_width
    if (nw, nh) != (w, h):
        out = bic(out, (nh - nh % 2, nw - nw % 2)).clamp(0, 1)
    return out
