"""Thin tensor-level wrappers over the iw3 entry points of libnunif_hip.so (no fallbacks)."""
import ctypes

import torch

from .. import _hip


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _cuda_f32(t, name):
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: tensor must live on a ROCm device (got {t.device}); there is no CPU fallback")
    return t.to(torch.float32).contiguous()


def resize_aa(x, size, mode="bilinear", align_corners=False, clamp01=False, mean=None, std=None):
    """F.interpolate(x, size, mode, align_corners, antialias=True) for [B,C,H,W] (+ optional fused clamp/normalise)."""
    assert mode in ("bilinear", "bicubic")
    x = _cuda_f32(x, "resize_aa")
    b, c, h, w = x.shape
    oh, ow = int(size[0]), int(size[1])
    y = torch.empty((b, c, oh, ow), dtype=torch.float32, device=x.device)
    tmp = torch.empty((b, c, h, ow), dtype=torch.float32, device=x.device)
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    s = (ctypes.c_float * 3)(*std) if std is not None else None
    if mean is not None:
        assert c == 3
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_resize_aa(_p(x), _p(y), _p(tmp), b * c, h, w, oh, ow,
                                                  1 if mode == "bicubic" else 0, 1 if align_corners else 0,
                                                  1 if clamp01 else 0, m, s, _hip.current_stream_ptr(x.device)))
    return y


def dilate_edge(x, n_x, n_y):
    x = _cuda_f32(x, "dilate_edge")
    b, c, h, w = x.shape
    assert c == 1
    y = torch.empty_like(x)
    work = torch.empty(_hip.lib().nunif_hip_dilate_edge_work_floats(b, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_dilate_edge(_p(x), _p(y), _p(work), b, h, w, n_x, n_y,
                                                    _hip.current_stream_ptr(x.device)))
    return y


def minmax_normalize(x):
    """Per-item (x-min)/(max-min), clamp [0,1]; x: [B,...]."""
    x = _cuda_f32(x, "minmax_normalize")
    b = x.shape[0]
    y = torch.empty_like(x)
    mm = torch.empty((b, 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_minmax_normalize(_p(x), _p(y), _p(mm), b, x[0].numel(),
                                                         _hip.current_stream_ptr(x.device)))
    return y


def minmax_keys(x):
    """Order-keyed (min, max) of every item of x [B, ...] -> [B, 2] fp32 device scratch (feeds ema_scaler_push)."""
    x = _cuda_f32(x, "minmax_keys")
    mm = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_minmax(_p(x), _p(mm), x.shape[0], x[0].numel(), _hip.current_stream_ptr(x.device)))
    return mm


def ema_scaler_push(state, keys, ring_size, count, filled, first, decay):
    with torch.cuda.device(state.device):
        _hip.check(_hip.lib().nunif_hip_ema_scaler_push(_p(state), _p(keys), ring_size, count, 1 if filled else 0,
                                                        1 if first else 0, float(decay), _hip.current_stream_ptr(state.device)))


def ema_scaler_ring_minmax(state, ring_size):
    with torch.cuda.device(state.device):
        _hip.check(_hip.lib().nunif_hip_ema_scaler_ring_minmax(_p(state), ring_size, _hip.current_stream_ptr(state.device)))


def range_normalize(x, lohi, max_mode=False):
    """clamp((x - lo) / (hi - lo), 0, 1) (or clamp(x / hi)) with the two extrema in a device tensor ``lohi`` [2]."""
    x = _cuda_f32(x, "range_normalize")
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_range_normalize(_p(x), _p(y), _p(lohi), x.numel(), 1 if max_mode else 0,
                                                        _hip.current_stream_ptr(x.device)))
    return y


def make_input_planes(depth, divergence_value, convergence_value, border_pix=0):
    """[B,1,H,W] depth -> [B,3,H,W] = depth | divergence plane | convergence plane (+ screen-border taper)."""
    depth = _cuda_f32(depth, "make_input_planes")
    b, _, h, w = depth.shape
    out = torch.empty((b, 3, h, w), dtype=torch.float32, device=depth.device)
    with torch.cuda.device(depth.device):
        _hip.check(_hip.lib().nunif_hip_make_input_planes(_p(depth), _p(out), b, h, w, float(divergence_value),
                                                          float(convergence_value), int(border_pix),
                                                          _hip.current_stream_ptr(depth.device)))
    return out


def stack(tensors):
    """torch.stack(tensors) for equally shaped contiguous device tensors, as ONE copy launch of the engine's own (a view
    when the tensors already are consecutive slices of one buffer)."""
    t0 = tensors[0]
    n = len(tensors)
    if not (t0.is_cuda and all(t.is_cuda and t.shape == t0.shape and t.dtype == t0.dtype and t.is_contiguous()
                               and t.device == t0.device for t in tensors)):
        return torch.stack(list(tensors))
    nbytes = t0.numel() * t0.element_size()
    if all(tensors[k].data_ptr() == t0.data_ptr() + k * nbytes and tensors[k].untyped_storage().data_ptr() ==
           t0.untyped_storage().data_ptr() for k in range(n)):
        return t0.as_strided((n, *t0.shape), (t0.numel(), *t0.stride()))          # consecutive slices: no copy at all
    out = torch.empty((n, *t0.shape), dtype=t0.dtype, device=t0.device)
    with torch.cuda.device(t0.device):
        for k0 in range(0, n, 16):
            grp = tensors[k0:k0 + 16]
            arr = (ctypes.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
            _hip.check(_hip.lib().nunif_hip_stack(arr, len(grp), nbytes, out[k0].data_ptr(),
                                                  _hip.current_stream_ptr(t0.device)))
    return out


VIEW = {"both": 0, "left": 1, "right": 2}


def forward_warp(c, depth, divergence, convergence, fill, synthetic_view, return_mask, width_base):
    c = _cuda_f32(c, "forward_warp")
    depth = _cuda_f32(depth, "forward_warp").to(c.device)
    b, ch, h, w = c.shape
    assert ch == 3 and depth.shape == (b, 1, h, w)
    view = VIEW[synthetic_view]
    left = torch.empty_like(c) if view != 2 else None
    right = torch.empty_like(c) if view != 1 else None
    lm = torch.empty((b, 1, h, w), dtype=torch.float32, device=c.device) if (return_mask and view != 2) else None
    rm = torch.empty((b, 1, h, w), dtype=torch.float32, device=c.device) if (return_mask and view != 1) else None
    p = _hip.ForwardWarpParams(b, h, w, float(divergence), float(convergence), 1 if fill else 0, view,
                               1 if width_base else 0)
    with torch.cuda.device(c.device):
        _hip.check(_hip.lib().nunif_hip_forward_warp(_p(c), _p(depth), _p(left), _p(right), _p(lm), _p(rm),
                                                     ctypes.byref(p), _hip.current_stream_ptr(c.device)))
    return left, right, lm, rm


def backward_warp(c, depth, divergence, convergence, synthetic_view):
    c = _cuda_f32(c, "backward_warp")
    depth = _cuda_f32(depth, "backward_warp").to(c.device)
    b, ch, h, w = c.shape
    dh, dw = depth.shape[2:]
    view = VIEW[synthetic_view]
    left = torch.empty_like(c) if view != 2 else None
    right = torch.empty_like(c) if view != 1 else None
    with torch.cuda.device(c.device):
        _hip.check(_hip.lib().nunif_hip_backward_warp(_p(c), _p(depth), _p(left), _p(right), b, ch, h, w, dh, dw,
                                                      float(divergence), float(convergence), view,
                                                      _hip.current_stream_ptr(c.device)))
    return left, right


def frame_to_tensor(frame_hwc, device=None):
    """uint8 / int16-held-uint16 HWC [H,W,3] -> CHW float on the device (VU.to_tensor, video.py:218-223).
    The frame is either a device tensor or a PINNED host tensor (zero-copy: the kernel reads it over PCIe; pass
    ``device``)."""
    if frame_hwc.device.type != "cuda":
        if not (frame_hwc.is_pinned() and device is not None):
            raise RuntimeError("frame_to_tensor: tensor must live on a ROCm device or in pinned host memory "
                               "(with device=...); there is no CPU fallback")
        dev = torch.device(device)
        assert frame_hwc.dim() == 3 and frame_hwc.shape[2] == 3 and frame_hwc.is_contiguous()
        bits = 8 if frame_hwc.dtype == torch.uint8 else 16
        h, w, _ = frame_hwc.shape
        out = torch.empty((3, h, w), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _hip.check(_hip.lib().nunif_hip_frame_to_tensor(_p(frame_hwc), _p(out), h, w, bits, _hip.current_stream_ptr(dev)))
        return out
    assert frame_hwc.dim() == 3 and frame_hwc.shape[2] == 3
    if frame_hwc.dtype == torch.uint8:
        bits, src = 8, frame_hwc.contiguous()
    elif frame_hwc.dtype in (torch.int16, torch.uint16):
        bits, src = 16, frame_hwc.contiguous()
    else:
        raise ValueError(f"unsupported frame dtype {frame_hwc.dtype}")
    h, w, _ = src.shape
    out = torch.empty((3, h, w), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _hip.check(_hip.lib().nunif_hip_frame_to_tensor(_p(src), _p(out), h, w, bits, _hip.current_stream_ptr(src.device)))
    return out


LAYOUT = {"sbs": 0, "cross_eyed": 1, "tb": 2}


def stereo_to_frame(left, right, layout="sbs", bits=8):
    """clamp(cat(left,right)) quantised to HWC uint8 (or uint16 stored as int16 bit pattern) in one pass."""
    left, right = _cuda_f32(left, "stereo_to_frame"), _cuda_f32(right, "stereo_to_frame")
    _, h, w = left.shape
    ho, wo = (2 * h, w) if layout == "tb" else (h, 2 * w)
    out = torch.empty((ho, wo, 3), dtype=torch.uint8 if bits == 8 else torch.int16, device=left.device)
    with torch.cuda.device(left.device):
        _hip.check(_hip.lib().nunif_hip_stereo_to_frame(_p(left), _p(right), _p(out), h, w, LAYOUT[layout], bits,
                                                        _hip.current_stream_ptr(left.device)))
    return out


def to_frame(x, bits=8, out=None):
    """CHW float -> clamp + quantise -> HWC uint8 (or uint16 bit pattern in int16): VU.to_frame (video.py:236-245).
    ``out`` may be a PINNED host tensor: the kernel then writes the frame straight into host memory (zero-copy)."""
    x = _cuda_f32(x, "to_frame")
    _, h, w = x.shape
    if out is None:
        out = torch.empty((h, w, 3), dtype=torch.uint8 if bits == 8 else torch.int16, device=x.device)
    else:
        assert out.shape == (h, w, 3) and out.is_contiguous() and (out.device.type == "cuda" or out.is_pinned())
        assert out.dtype == (torch.uint8 if bits == 8 else torch.int16)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_stereo_to_frame(_p(x), None, _p(out), h, w, 3, bits,
                                                        _hip.current_stream_ptr(x.device)))
    return out


def stereo_compose(left, right, layout="sbs"):
    left, right = _cuda_f32(left, "stereo_compose"), _cuda_f32(right, "stereo_compose")
    _, h, w = left.shape
    ho, wo = (2 * h, w) if layout == "tb" else (h, 2 * w)
    out = torch.empty((3, ho, wo), dtype=torch.float32, device=left.device)
    with torch.cuda.device(left.device):
        _hip.check(_hip.lib().nunif_hip_stereo_compose(_p(left), _p(right), _p(out), h, w, LAYOUT[layout],
                                                       _hip.current_stream_ptr(left.device)))
    return out


def map_depth(x, kind, p0=0.0, p1=0.0):
    x = _cuda_f32(x, "map_depth")
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_map_depth(_p(x), _p(y), x.numel(), kind, float(p0), float(p1),
                                                  _hip.current_stream_ptr(x.device)))
    return y


def delta_warp(c, delta, delta_scale, flip=False):
    """backward_warp(c, make_grid, [delta, 0], delta_scale) (+ the right eye's mirror) — iw3/backward_warp.py:67-93."""
    c = _cuda_f32(c, "delta_warp")
    delta = _cuda_f32(delta, "delta_warp")
    b, ch, h, w = c.shape
    assert delta.shape[0] == b and delta.shape[1] == 1
    out = torch.empty_like(c)
    with torch.cuda.device(c.device):
        _hip.check(_hip.lib().nunif_hip_delta_warp(_p(c), _p(delta), _p(out), b, ch, h, w, delta.shape[2], delta.shape[3],
                                                   float(delta_scale), 1 if flip else 0,
                                                   _hip.current_stream_ptr(c.device)))
    return out


def delta_weight_warp(c, delta, weight, delta_scale, flip=False):
    """clamp(sum_i backward_warp(c, delta_i) * weight_i) — iw3/backward_warp.py:300-321; weight at image resolution."""
    c, delta, weight = _cuda_f32(c, "delta_weight_warp"), _cuda_f32(delta, "delta_weight_warp"), _cuda_f32(weight, "delta_weight_warp")
    b, ch, h, w = c.shape
    layers = delta.shape[1]
    assert weight.shape == (b, layers, h, w)
    out = torch.empty_like(c)
    with torch.cuda.device(c.device):
        _hip.check(_hip.lib().nunif_hip_delta_weight_warp(_p(c), _p(delta), _p(weight), _p(out), b, ch, h, w, delta.shape[2],
                                                          delta.shape[3], layers, float(delta_scale), 1 if flip else 0,
                                                          _hip.current_stream_ptr(c.device)))
    return out


def hole_mask_postprocess(mask_logits, target_size, threshold, inner_iter=0, outer_iter=0, z=None):
    """``nunif_hip_hole_mask_postprocess``: closing + bilinear resize + sigmoid > threshold + horizontal OR-dilations.
    mask_logits [B,1,h,w] f32 -> bool [B,1,H,W]; ``z`` ([B,C,H,W] f32, optional) is zeroed in place where the mask is set."""
    x = _cuda_f32(mask_logits, "hole_mask_postprocess")
    B, one, h, w = x.shape
    assert one == 1
    H, W = int(target_size[0]), int(target_size[1])
    mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=x.device)
    work = torch.empty((2 * B * h * w + (B * H * W + 3) // 4,), dtype=torch.float32, device=x.device)
    if z is not None:
        assert z.dtype == torch.float32 and z.is_contiguous() and z.shape[0] == B and tuple(z.shape[2:]) == (H, W)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_hole_mask_postprocess(
            _p(x), _p(mask), _p(work), B, h, w, H, W, float(threshold), int(inner_iter), int(outer_iter),
            _p(z), int(z.shape[1]) if z is not None else 0,
            _hip.current_stream_ptr(x.device)))
    return mask.bool()


def anaglyph(left, right, mode):
    left, right = _cuda_f32(left, "anaglyph"), _cuda_f32(right, "anaglyph")
    assert left.ndim == 3 and left.shape[0] == 3 and left.shape == right.shape
    out = torch.empty_like(left)
    with torch.cuda.device(left.device):
        _hip.check(_hip.lib().nunif_hip_anaglyph(_p(left), _p(right), _p(out), left.shape[1], left.shape[2], int(mode),
                                                 _hip.current_stream_ptr(left.device)))
    return out


def equirectangular(c):
    c = _cuda_f32(c, "equirectangular")
    assert c.ndim == 3
    C, h, w = c.shape
    max_edge = max(h, w)
    size = max_edge + max_edge // 2
    Hp, Wp = h + 2 * ((size - h) // 2), w + 2 * ((size - w) // 2)
    out = torch.empty((C, Hp, Wp), dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        _hip.check(_hip.lib().nunif_hip_equirectangular(_p(c), _p(out), C, h, w, _hip.current_stream_ptr(c.device)))
    return out


def reflection_pad2d(x, padding):
    """``reflection_pad2d_naive`` (nunif/modules/reflection_pad2d.py:13-48): (left, right, top, bottom); positive = reflect
    (edge pixel not repeated), negative = crop.  x: [B,C,H,W] fp32."""
    x = _cuda_f32(x, "reflection_pad2d")
    assert x.dim() == 4 and len(padding) == 4
    left, right, top, bottom = (int(p) for p in padding)
    b, c, h, w = x.shape
    y = torch.empty((b, c, h + top + bottom, w + left + right), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_reflection_pad2d(_p(x), _p(y), b * c, h, w, left, right, top, bottom,
                                                         _hip.current_stream_ptr(x.device)))
    return y


def depth_postprocess(x, max_dist=None, to_disparity=False, eps=0.1, negate=False):
    """nan_to_num -> clamp(max=max_dist) -> 1/(d+eps) -> optional sign flip (video_depth_anything_model.py:66-76,88-90)."""
    x = _cuda_f32(x, "depth_postprocess")
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_depth_postprocess(_p(x), _p(y), x.numel(), float(max_dist) if max_dist else 0.0,
                                                          int(bool(to_disparity)), float(eps), int(bool(negate)),
                                                          _hip.current_stream_ptr(x.device)))
    return y


def mask_morphology(mask, op, n_a, n_b=0):
    """Stand-alone mask morphology (see ``nunif_hip_mask_morphology``).  mask: [B,1,H,W] (bool / float) -> fp32 0/1."""
    assert mask.dim() == 4 and mask.shape[1] == 1
    if mask.device.type != "cuda":
        raise RuntimeError("mask_morphology: expected a ROCm tensor; there is no CPU path")
    x = mask.to(torch.float32).contiguous()
    b, _, h, w = x.shape
    y = torch.empty_like(x)
    work = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _hip.check(_hip.lib().nunif_hip_mask_morphology(_p(x), _p(y), _p(work), b, h, w, op, int(n_a), int(n_b),
                                                        _hip.current_stream_ptr(x.device)))
    return y
