"""8-way dihedral test-time augmentation on the HIP engine.  Mirrors ``nunif/transforms/tta.py`` ``tta_split`` :20-34 /
``tta_merge`` :37-48: ``nunif_hip_tta_view`` writes each view in one pass, ``nunif_hip_tta_merge`` reads the eight model
outputs once, undoes the transforms on the fly, adds them in the reference's order (bit-exact), scales and clamps."""
import ctypes

import torch

from ... import _hip


def _f32(x, name):
    if x.device.type != "cuda":
        raise RuntimeError(f"{name}: tensor must live on a ROCm device (got {x.device}); there is no CPU fallback")
    return x.to(torch.float32).contiguous()


def tta_split(x):
    assert isinstance(x, torch.Tensor) and x.dim() == 3
    dtype = x.dtype
    x = _f32(x, "tta_split")
    c, h, w = x.shape
    views = [x]
    with torch.cuda.device(x.device):
        for v in range(1, 8):
            y = torch.empty((c, w, h) if v & 4 else (c, h, w), dtype=torch.float32, device=x.device)
            _hip.check(_hip.lib().nunif_hip_tta_view(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), c, h, w, v,
                                                     _hip.current_stream_ptr(x.device)))
            views.append(y)
    return tuple(v.to(dtype) for v in views)


def tta_merge(xs):
    assert len(xs) == 8
    dtype = xs[0].dtype
    xs = [_f32(x, "tta_merge") for x in xs]
    c, h, w = xs[0].shape
    for k, x in enumerate(xs):
        assert tuple(x.shape) == ((c, w, h) if k & 4 else (c, h, w)), f"view {k} has shape {tuple(x.shape)}"
    out = torch.empty((c, h, w), dtype=torch.float32, device=xs[0].device)
    ptrs = (ctypes.c_void_p * 8)(*[x.data_ptr() for x in xs])
    with torch.cuda.device(out.device):
        _hip.check(_hip.lib().nunif_hip_tta_merge(ptrs, ctypes.c_void_p(out.data_ptr()), c, h, w,
                                                  _hip.current_stream_ptr(out.device)))
    return out.to(dtype)
