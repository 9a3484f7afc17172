"""``inpaint.light_inpaint_v1`` on the HIP engine.

Mirrors ``iw3/models/light_inpaint_v1.py`` ``LightInpaintV1`` :53-161: registry name, i2i geometry (scale 1, offset 16,
blend 8), the ``state_dict`` key layout and ``infer(x, mask, closing, inner_dilation, outer_dilation, base_width)`` :106-110
(= ``preprocess`` :93-104 + ``forward(..., skip_i2i_offset=True)``), which is what ``MLBWInpaintImage`` calls.  The whole
of it is one C call, ``nunif_hip_light_inpaint_infer`` (nunif_amd/csrc/light_inpaint.hip).  The training-style
``forward(x, soft_mask)`` entry is not provided.
"""
import ctypes
import math
from collections import OrderedDict

import torch

from ...nunif.models import I2IBaseModel, register_model
from ... import _hip

OFFSET = 16


def _init_weights():
    """Fresh weights in the reference's key layout (basic_module_init-style scales; proj_spatial ~ 0 with bias 1)."""
    sd = OrderedDict()

    def lin(key, *shape):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        sd[key + ".weight"] = torch.randn(shape) * math.sqrt(1.0 / fan_in)
        sd[key + ".bias"] = torch.zeros(shape[0])

    def block(p, C, ws):
        N = ws * ws
        lin(p + "gmlp.gmlp.proj_in", 4 * C, C)
        sd[p + "gmlp.gmlp.proj_spatial.weight"] = (torch.rand(N, N, 1) * 2 - 1) * (1e-3 / C)
        sd[p + "gmlp.gmlp.proj_spatial.bias"] = torch.ones(N)
        lin(p + "gmlp.gmlp.proj_out", C, 2 * C)
        sd[p + "norm1.weight"] = torch.ones(C)
        sd[p + "norm2.weight"] = torch.ones(2 * C)
        lin(p + "glu_conv.w1", C, C, 1, 1)
        lin(p + "glu_conv.w2", C, C // 2, 3, 3)

    sd["mask_bias"] = torch.randn(1, 96, 1, 1) * 0.01
    lin("patch.0", 96, 48, 1, 1)
    block("enc1.", 96, 16)
    lin("down", 192, 96, 2, 2)
    for i in range(4):
        block(f"enc2.{i}.", 192, 8)
    lin("up", 384, 192, 1, 1)
    block("dec1.", 96, 16)
    lin("to_image.1", 48, 96, 3, 3)
    return sd


class HipLightInpaintEngine:
    def __init__(self, state_dict, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the light_inpaint HIP engine needs a ROCm device (model.to('cuda:N')); no CPU fallback")
        keep, descs = [], []
        for name, t in state_dict.items():
            t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            keep.append(t)
            d = _hip.TensorDesc()
            d.name, d.data, d.ndim = name.encode(), t.data_ptr(), t.dim()
            for i, s in enumerate(t.shape):
                d.shape[i] = s
            descs.append(d)
        arr = (_hip.TensorDesc * len(descs))(*descs)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_light_inpaint_create(arr, len(descs), ctypes.byref(handle)))
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().nunif_hip_light_inpaint_destroy(h)
            except Exception:
                pass

    def infer(self, x, mask, closing, inner_iter, outer_iter, mirror_x=False):
        B, C, H, W = x.shape
        assert C == 3 and tuple(mask.shape) == (B, 1, H, W)
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_light_inpaint_infer_ex(
                self.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                B, H, W, 1 if closing else 0, int(inner_iter), int(outer_iter), 1 if mirror_x else 0,
                _hip.current_stream_ptr(self.device)))
        return out


@register_model
class LightInpaintV1(I2IBaseModel):
    name = "inpaint.light_inpaint_v1"

    def __init__(self):
        super().__init__({}, scale=1, offset=OFFSET, in_channels=3, blend_size=8)
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self.downscaling_factor, self.mod = 4, 16
        self._weights = _init_weights()
        self._engine = None

    def get_device(self):
        return self._device_probe.device

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._weights.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        missing = [k for k in self._weights if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._weights]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for LightInpaintV1: missing {missing[:4]}, "
                               f"unexpected {unexpected[:4]}")
        for k in self._weights:
            if k in state_dict:
                v = state_dict[k].detach().to("cpu")
                if v.shape != self._weights[k].shape:
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._weights[k].shape)}")
                self._weights[k] = v.float().clone()
        self._engine = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def parameters(self, recurse=True):
        return iter(self._weights.values())

    def half(self):
        return self

    def float(self):
        return self

    def engine(self):
        dev = self.get_device()
        if self._engine is None or self._engine.device != dev:
            self._engine = HipLightInpaintEngine(self._weights, dev)
        return self._engine

    supports_mirror_x = True

    def infer(self, x, mask, closing=False, inner_dilation=0, outer_dilation=0, base_width=None, mirror_x=False):
        """x [B,3,H,W] float in [0,1], mask [B,1,H,W] bool (True = hole) -> inpainted [B,3,H,W].
        ``mirror_x`` (not in the reference): ``flip(infer(flip(x), mask))`` with ``mask`` given in the flipped frame, no flip passes."""
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        dev = self.get_device()
        W = x.shape[-1]

        def n_iter(n):          # dilate_inner / dilate_outer (iw3/dilation.py:74-103)
            if n <= 0:
                return 0
            return max(round(W / base_width * n), 1) if base_width is not None else n
        m = mask.to(device=dev)
        # the engine takes a byte per pixel, hole = non-zero: a bool mask IS that (zero-copy view), a uint8 mask passes as it is
        if m.dtype == torch.bool:
            m = m.contiguous().view(torch.uint8)
        elif m.dtype != torch.uint8:
            m = (m > 0).to(torch.uint8)
        m = m.contiguous()
        dtype = x.dtype
        out = self.engine().infer(x.to(device=dev, dtype=torch.float32).contiguous(), m, closing,
                                  n_iter(inner_dilation), n_iter(outer_dilation), mirror_x=mirror_x)
        return out.to(dtype)

    def forward(self, x, mask, skip_i2i_offset=False):
        raise NotImplementedError("the HIP engine implements LightInpaintV1.infer (hard hole mask in, inpainted frame out); "
                                  "the training-style forward(x, soft_mask) is not provided")
