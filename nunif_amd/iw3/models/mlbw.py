"""iw3 ``sbs.mlbw`` (multi-layer backward warp) on the HIP engine.

Mirrors ``iw3/models/mlbw.py`` (reference) ``MLBW`` :37-247 — registry name + the factories ``sbs.mlbw_l2`` /
``sbs.mlbw_l4`` / ``sbs.mlbw_l2s`` / ``sbs.mlbw_l4s`` :269-284, constructor kwargs, ``i2i_*`` attributes (scale 1,
offset 32, in_channels 8, blend_size 4), ``num_layers`` / ``delta_output`` / ``symmetric`` / ``hole_mask`` attributes and
the ``state_dict`` key layout (``lv1_in.1``, ``lv2.N.*``, ``lv1_out.1``).  Only the inference path iw3 uses is on the
engine: ``delta_output = True`` (``_forward_delta_only`` :238-247) -> ``(delta, layer_weight)``, consumed by
``apply_divergence_nn_delta_weight``.  ``hole_mask=True`` (``sbs.mask_mlbw_l2`` :275-278) adds the hole-logit channel
(``_forward`` :104-106, ``_forward_delta_only`` :234-238).  The ``cycle`` (training) variant is not provided.
"""
import ctypes
import math
from collections import OrderedDict

import torch

from ...nunif.models import I2IBaseModel, register_model, register_model_factory
from ... import _hip
from .row_flow_v3 import _score_bias_input

OFFSET = 32


def _init_weights(num_layers, small, hole_mask=False):
    C = 32 * num_layers
    sd = OrderedDict()

    def lin(key, *shape):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        sd[key + ".weight"] = torch.randn(shape) * math.sqrt(1.0 / fan_in)
        sd[key + ".bias"] = torch.zeros(shape[0])

    lin("lv1_in.1", C // 8, 3, 1, 9)
    for i in range(2 if small else 4):
        p = f"lv2.{i}."
        lin(p + "mha.mha.qkv_proj", 3 * C, C)
        lin(p + "mha.mha.head_proj", C, C)
        lin(p + "conv_mlp.0", C, C, 1, 1)
        lin(p + "conv_mlp.3", C, C, 3, 3)
        sd[p + "bias.index"], sd[p + "bias.delta"] = _score_bias_input((4, 4))
        lin(p + "bias.to_bias.0", 8, 2)
        lin(p + "bias.to_bias.2", 1, 8)
    lin("lv1_out.1", 2 * num_layers + (1 if hole_mask else 0), C // 8, 1, 9)
    return sd


class HipMLBWEngine:
    def __init__(self, state_dict, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the mlbw HIP engine needs a ROCm device (model.to('cuda:N')); no CPU fallback")
        keep, descs = [], []
        for name, t in state_dict.items():
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
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_mlbw_create(arr, len(descs), ctypes.byref(handle)))
        self.handle = handle
        self.num_layers = _hip.lib().nunif_hip_mlbw_num_layers(handle)
        self.hole_mask = bool(_hip.lib().nunif_hip_mlbw_has_hole_mask(handle))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().nunif_hip_mlbw_destroy(h)
            except Exception:
                pass

    def delta(self, x, flip=False):
        B, C, h, w = x.shape
        assert C == 3
        delta = torch.empty((B, self.num_layers, h, w), dtype=torch.float32, device=self.device)
        weight = torch.empty_like(delta)
        mask = torch.empty((B, 1, h, w), dtype=torch.float32, device=self.device) if self.hole_mask else None
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_mlbw_delta_mask(
                self.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(delta.data_ptr()),
                ctypes.c_void_p(weight.data_ptr()), ctypes.c_void_p(mask.data_ptr() if mask is not None else None),
                B, h, w, 1 if flip else 0, _hip.current_stream_ptr(self.device)))
        if self.hole_mask:
            return delta, weight, mask          # mask logits are already flipped back to image coordinates
        return delta, weight


@register_model
class MLBW(I2IBaseModel):
    name = "sbs.mlbw"

    def __init__(self, num_layers=2, base_dim=32, small=False, cycle=False, hole_mask=False, **kwargs):
        super().__init__(dict(num_layers=num_layers, base_dim=base_dim, small=small, cycle=cycle, hole_mask=hole_mask),
                         scale=1, offset=OFFSET, in_channels=8, blend_size=4)
        if base_dim != 32 or num_layers not in (2, 4):
            raise ValueError("the HIP mlbw engine supports base_dim = 32 with 2 or 4 layers (heads of 32 channels)")
        if cycle:
            raise NotImplementedError("the cycle (training) MLBW variant is not on the HIP engine")
        if hole_mask and num_layers != 2:
            raise ValueError("the hole-mask variant is registered for 2 layers only (sbs.mask_mlbw_l2)")
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self.num_layers = num_layers
        self.cycle, self.hole_mask = cycle, hole_mask
        self.delta_output = False
        self.symmetric = False
        self._weights = _init_weights(num_layers, small, hole_mask)
        self._engine = None

    def get_device(self):
        return self._device_probe.device

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._weights.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        missing = [k for k in self._weights if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._weights]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for MLBW: missing {missing[:4]}, unexpected {unexpected[:4]}")
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
            self._engine = HipMLBWEngine(self._weights, dev)
        return self._engine

    def infer_delta(self, x, flip=False):
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        return self.engine().delta(x.to(device=self.get_device(), dtype=torch.float32).contiguous(), flip=flip)

    def forward(self, x):
        if not self.delta_output:
            raise NotImplementedError("the HIP engine implements the delta_output path (what iw3 inference uses); "
                                      "set model.delta_output = True")
        if x.shape[1] == 8:
            x = x[:, 3:6]
        return self.infer_delta(x)


register_model_factory("sbs.mlbw_l2", lambda **kwargs: MLBW(num_layers=2, base_dim=32, **kwargs))
register_model_factory("sbs.mlbw_l4", lambda **kwargs: MLBW(num_layers=4, base_dim=32, **kwargs))
register_model_factory("sbs.mask_mlbw_l2", lambda **kwargs: MLBW(num_layers=2, base_dim=32, hole_mask=True, **kwargs))
register_model_factory("sbs.mlbw_l2s", lambda **kwargs: MLBW(num_layers=2, base_dim=32, small=True, **kwargs))
register_model_factory("sbs.mlbw_l4s", lambda **kwargs: MLBW(num_layers=4, base_dim=32, small=True, **kwargs))
