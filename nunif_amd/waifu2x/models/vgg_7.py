"""waifu2x VGG7 / UpConv7 on the HIP engine.

Mirrors ``waifu2x/models/vgg_7.py`` ``VGG7`` :6-37 (seven 3x3 VALID convolutions 3-32-32-64-64-128-128-3, LeakyReLU(0.1),
scale 1, offset 7) and ``waifu2x/models/upconv_7.py`` ``UpConv7`` :6-42 (3-16-32-64-128-128-256 + ConvTranspose2d(256, 3, 4,
2, 3), scale 2, offset 14): registry names, constructor kwargs, ``i2i_*`` geometry, the ``net.N`` state-dict keys and the
eval clamp.  The forward pass is the conv-stack branch of ``nunif_hip_cunet_forward`` (nunif_amd/csrc/cunet.cpp
``forward_stack``): first layer on the VALU, the rest on the implicit-GEMM conv kernel, the transposed conv as a
2x2-window gather GEMM with a pixel-shuffle store.
"""
import math
from collections import OrderedDict

import torch

from ...nunif.models import I2IBaseModel, register_model
from .cunet import HipCUNetEngine


class _ConvStack(I2IBaseModel):
    _channels = ()
    _deconv = False
    _kaiming = False

    def __init__(self, in_channels=3, out_channels=3, *, scale, offset):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels), scale=scale, offset=offset,
                         in_channels=in_channels)
        if in_channels != 3 or out_channels != 3:
            raise ValueError("the HIP conv-stack engine supports in_channels = out_channels = 3")
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self._weights = self._init_weights()
        self._engine = None

    def _init_weights(self):
        sd = OrderedDict()
        ch = self._channels
        n = len(ch) - 1
        for i in range(n):
            cin, cout = ch[i], ch[i + 1]
            last_deconv = self._deconv and i == n - 1
            k = 4 if last_deconv else 3
            shape = (cin, cout, k, k) if last_deconv else (cout, cin, k, k)
            if self._kaiming:          # upconv_7.py:25-29: kaiming_normal_(fan_out, relu), zero bias
                fan_out = shape[0] * k * k
                w = torch.randn(shape) * math.sqrt(2.0 / fan_out)
                b = torch.zeros(cout)
            else:                      # nn.Conv2d default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for both
                bound = 1.0 / math.sqrt(cin * k * k)
                w = (torch.rand(shape) * 2 - 1) * bound
                b = (torch.rand(cout) * 2 - 1) * bound
            sd[f"net.{2 * i}.weight"], sd[f"net.{2 * i}.bias"] = w, b
        return sd

    def get_device(self):
        return self._device_probe.device

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._weights.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        missing = [k for k in self._weights if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._weights]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing {missing[:4]}, "
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
            self._engine = HipCUNetEngine(self._weights, False, dev, scale=self.i2i_scale, offset=self.i2i_offset)
        return self._engine

    def forward(self, x):
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        dtype = x.dtype
        return self.engine().forward(x.to(device=self.get_device(), dtype=torch.float32).contiguous()).to(dtype)

    def render_frame(self, x, tile_size, batch_size):
        return self.engine().render(x.to(device=self.get_device(), dtype=torch.float32).contiguous(), tile_size, batch_size)


@register_model
class VGG7(_ConvStack):
    name = "waifu2x.vgg_7"
    _channels = (3, 32, 32, 64, 64, 128, 128, 3)

    def __init__(self, in_channels=3, out_channels=3):
        super().__init__(in_channels, out_channels, scale=1, offset=7)


@register_model
class UpConv7(_ConvStack):
    name = "waifu2x.upconv_7"
    _channels = (3, 16, 32, 64, 128, 128, 256, 3)
    _deconv = True
    _kaiming = True

    def __init__(self, in_channels=3, out_channels=3):
        super().__init__(in_channels, out_channels, scale=2, offset=14)
