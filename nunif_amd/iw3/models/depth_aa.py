"""iw3 ``iw3.depth_aa`` (depth anti-aliasing) on the HIP engine.

Mirrors ``iw3/models/depth_aa.py`` (reference) ``DepthAA`` :29-95 — registry name, ``i2i_*`` attributes (scale 1,
offset 0, in_channels 1, blend_size 0), ``infer(x)`` :46-56, ``forward(x, clamp=None)`` :59-85 and the ``state_dict``
key layout (``proj_in``, ``blocks.N.*``, ``proj_out``), so ``iw3_depth_aa_20250530.pth`` loads unchanged.  The net is
``nunif_hip_depth_aa_forward`` (nunif_amd/csrc/depth_aa.hip).
"""
import ctypes
import math
from collections import OrderedDict

import torch

from ...nunif.models import I2IBaseModel, register_model
from ... import _hip
from .row_flow_v3 import _score_bias_input


def _init_weights():
    sd = OrderedDict()

    def lin(key, *shape, zero=False):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        sd[key + ".weight"] = torch.zeros(shape) if zero else torch.randn(shape) * math.sqrt(1.0 / fan_in)
        sd[key + ".bias"] = torch.zeros(shape[0])

    lin("proj_in", 32, 4, 1, 1)
    for i in range(3):
        p = f"blocks.{i}."
        lin(p + "mha.mha.qkv_proj", 96, 32)
        lin(p + "mha.mha.head_proj", 32, 32)
        lin(p + "conv_mlp.0", 32, 32, 1, 1)
        lin(p + "conv_mlp.3", 32, 32, 3, 3)
        sd[p + "bias.index"], sd[p + "bias.delta"] = _score_bias_input((8, 8))
        lin(p + "bias.to_bias.0", 16, 2)
        lin(p + "bias.to_bias.2", 1, 16)
    lin("proj_out", 4, 32, 1, 1, zero=True)           # nn.init.constant_(proj_out.weight, 0)  (depth_aa.py:43)
    return sd


@register_model
class DepthAA(I2IBaseModel):
    name = "iw3.depth_aa"

    def __init__(self):
        super().__init__({}, scale=1, offset=0, in_channels=1, blend_size=0)
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self._weights = _init_weights()
        self._handle = None
        self._handle_device = None

    def get_device(self):
        return self._device_probe.device

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._weights.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        missing = [k for k in self._weights if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._weights]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for DepthAA: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for k in self._weights:
            if k in state_dict:
                v = state_dict[k].detach().to("cpu")
                if v.shape != self._weights[k].shape:
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._weights[k].shape)}")
                self._weights[k] = v.clone() if not torch.is_floating_point(v) else v.float().clone()
        self._release()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def parameters(self, recurse=True):
        return iter(v for v in self._weights.values() if torch.is_floating_point(v))

    def half(self):
        return self

    def float(self):
        return self

    def _release(self):
        h, self._handle = self._handle, None
        if h:
            try:
                _hip.lib().nunif_hip_depth_aa_destroy(h)
            except Exception:
                pass

    def __del__(self):
        self._release()

    def _engine(self):
        dev = self.get_device()
        if dev.type != "cuda":
            raise RuntimeError("the depth_aa HIP engine needs a ROCm device (model.to('cuda:N')); no CPU fallback")
        if self._handle is None or self._handle_device != dev:
            self._release()
            keep, descs = [], []
            for name, t in self._weights.items():
                if not torch.is_floating_point(t):
                    continue
                t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
                keep.append(t)
                d = _hip.TensorDesc()
                d.name, d.data, d.ndim = name.encode(), t.data_ptr(), t.dim()
                for i, s in enumerate(t.shape):
                    d.shape[i] = s
                descs.append(d)
            arr = (_hip.TensorDesc * len(descs))(*descs)
            handle = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _hip.check(_hip.lib().nunif_hip_depth_aa_create(arr, len(descs), ctypes.byref(handle)))
            self._handle, self._handle_device = handle, dev
        return self._handle

    def _run(self, x, mode):
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        squeeze = x.ndim == 3
        if squeeze:
            x = x.unsqueeze(0)
        dev = self.get_device()
        handle = self._engine()
        xin = x.to(device=dev, dtype=torch.float32).contiguous()
        B, C, h, w = xin.shape
        assert C == 1
        y = torch.empty_like(xin)
        with torch.cuda.device(dev):
            _hip.check(_hip.lib().nunif_hip_depth_aa_forward(handle, ctypes.c_void_p(xin.data_ptr()),
                                                             ctypes.c_void_p(y.data_ptr()), B, h, w, mode,
                                                             _hip.current_stream_ptr(dev)))
        y = y.to(x.dtype)
        return y.squeeze(0) if squeeze else y

    @torch.inference_mode()
    def infer(self, x):
        return self._run(x, 2)

    def forward(self, x, clamp=None):
        if clamp is None:
            clamp = True                                  # eval (the only mode the engine has)
        return self._run(x, 1 if clamp else 0)

    def load(self):
        raise RuntimeError("no network access: load the released state dict with nunif.models.load_model / load_state_dict")
