"""iw3 ``sbs.row_flow_v3`` on the HIP engine.

Mirrors ``iw3/models/row_flow_v3.py`` (reference) ``RowFlowV3`` :33-107 — registry name, ``i2i_*`` attributes
(scale 1, offset 32, in_channels 8, blend_size 4), the ``delta_output`` / ``symmetric`` switches, ``delta_scale`` buffer
and the ``state_dict`` key layout (``blocks.*``, ``last_layer.1.*``), so the released ``iw3_row_flow_v3_*.pth`` loads
unchanged.  Only the inference path the CLI uses is on the engine: ``delta_output = True`` (``_forward_delta_only``
:102-107), called by ``apply_divergence_nn_delta`` (iw3/backward_warp.py:191-236).  The net itself is
``nunif_hip_row_flow_delta`` (nunif_amd/csrc/rowflow.hip).
"""
import ctypes
import math
from collections import OrderedDict

import torch

from ...nunif.models import I2IBaseModel, register_model
from ... import _hip

OFFSET = 32


def _score_bias_input(window):
    """WindowScoreBias buffers (nunif/modules/attention.py:347-372)."""
    pos = [(y, x) for y in range(window[0]) for x in range(window[1])]
    delta = [(a[0] - b[0], a[1] - b[1]) for a in pos for b in pos]
    uniq = sorted(set(delta))
    index = torch.tensor([uniq.index(d) for d in delta], dtype=torch.int64)
    ud = torch.tensor(uniq, dtype=torch.float32)
    return index, ud / ud.abs().max()


def _init_weights():
    sd = OrderedDict()

    def lin(key, *shape, bias_n):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        sd[key + ".weight"] = torch.randn(shape) * math.sqrt(1.0 / fan_in)
        sd[key + ".bias"] = torch.zeros(bias_n)

    sd["delta_scale"] = torch.tensor(1.0 / 127.0)
    lin("blocks.0", 64, 24, 1, 1, bias_n=64)
    for bi, window in ((1, (4, 4)), (2, (3, 3))):
        p = f"blocks.{bi}."
        hidden = int((window[0] * window[1]) ** 0.5) * 2
        lin(p + "mha.mha.qkv_proj", 192, 64, bias_n=192)
        lin(p + "mha.mha.head_proj", 64, 64, bias_n=64)
        lin(p + "conv_mlp.0", 64, 64, 1, 1, bias_n=64)
        lin(p + "conv_mlp.3", 64, 64, 3, 3, bias_n=64)
        sd[p + "bias.index"], sd[p + "bias.delta"] = _score_bias_input(window)
        lin(p + "bias.to_bias.0", hidden, 2, bias_n=hidden)
        lin(p + "bias.to_bias.2", 1, hidden, bias_n=1)
    lin("last_layer.1", 1, 8, 3, 3, bias_n=1)
    return sd


class HipRowFlowEngine:
    def __init__(self, state_dict, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the row_flow_v3 HIP engine needs a ROCm device (model.to('cuda:N')); no CPU fallback")
        keep, descs = [], []
        for name, t in state_dict.items():
            if not torch.is_floating_point(t):
                continue                                   # bias.index (int64) is rebuilt from the window size
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
            _hip.check(_hip.lib().nunif_hip_row_flow_create(arr, len(descs), ctypes.byref(handle)))
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().nunif_hip_row_flow_destroy(h)
            except Exception:
                pass

    def delta(self, x, flip=False):
        B, C, h, w = x.shape
        assert C == 3
        out = torch.empty((B, 1, h, w), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_row_flow_delta(self.handle, ctypes.c_void_p(x.data_ptr()),
                                                           ctypes.c_void_p(out.data_ptr()), B, h, w, 1 if flip else 0,
                                                           _hip.current_stream_ptr(self.device)))
        return out


@register_model
class RowFlowV3(I2IBaseModel):
    name = "sbs.row_flow_v3"

    def __init__(self):
        super().__init__({}, scale=1, offset=OFFSET, in_channels=8, blend_size=4)
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self._weights = _init_weights()
        self._engine = None
        self.delta_output = False
        self.symmetric = False

    @property
    def delta_scale(self):
        return self._weights["delta_scale"]

    def get_device(self):
        return self._device_probe.device

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._weights.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        missing = [k for k in self._weights if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._weights]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for RowFlowV3: missing {missing[:4]}, "
                               f"unexpected {unexpected[:4]}")
        for k in self._weights:
            if k in state_dict:
                v = state_dict[k].detach().to("cpu")
                if v.shape != self._weights[k].shape:
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._weights[k].shape)}")
                self._weights[k] = v.clone() if not torch.is_floating_point(v) else v.float().clone()
        self._engine = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def parameters(self, recurse=True):
        return iter(v for v in self._weights.values() if torch.is_floating_point(v))

    def half(self):
        return self

    def float(self):
        return self

    def engine(self):
        dev = self.get_device()
        if self._engine is None or self._engine.device != dev:
            self._engine = HipRowFlowEngine(self._weights, dev)
        return self._engine

    def infer_delta(self, x, flip=False):
        """[B,3,h,w] feature planes -> horizontal flow [B,1,h,w]; ``flip`` mirrors the planes inside the first kernel."""
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        return self.engine().delta(x.to(device=self.get_device(), dtype=torch.float32).contiguous(), flip=flip)

    def forward(self, x):
        if not self.delta_output:
            raise NotImplementedError("the HIP engine implements the delta_output path (what iw3 inference uses); "
                                      "set model.delta_output = True")
        # (symmetric only changes what the CALLER does with the flow: +delta for the left eye, -delta for the right)
        if x.shape[1] == 8:                     # training-style packed input: rgb | depth feat | grid
            x = x[:, 3:6]
        delta = self.infer_delta(x)
        return torch.cat([delta, torch.zeros_like(delta)], dim=1)            # _forward_delta_only :102-107
