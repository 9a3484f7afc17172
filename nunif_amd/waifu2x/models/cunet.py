"""waifu2x CUNet / UpCUNet on the HIP engine.

Mirrors ``waifu2x/models/cunet.py`` (reference) ``CUNet`` :172-203 and ``UpCUNet`` :139-169 — registry names,
constructor kwargs, ``i2i_*`` geometry (scale 1 / offset 28 and scale 2 / offset 36, no blending),
``tile_size_validator`` :124-125 and the ``state_dict`` key layout, so reference ``.pth`` files load unchanged.  The
forward pass is ``nunif_hip_cunet_forward`` (nunif_amd/csrc/cunet.cpp); which net it runs is decided by the shape of
``unet1.conv_bottom.weight`` (a 4x4 ConvTranspose2d for UpCUNet).
"""
import ctypes
import math
from collections import OrderedDict

import torch

from ...nunif.models import I2IBaseModel, register_model
from ... import _hip


def tile_size_validator(size):
    return size % 4 == 0


def _init_weights(in_channels, out_channels, up=False):
    """Fresh kaiming-normal(fan_out) weights, zero biases, in the reference's key layout (cunet.py:43-50,89-96)."""
    sd = OrderedDict()

    def conv(key, cin, cout, k, transposed=False):
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        fan_out = (cin if transposed else cout) * k * k      # torch's fan_out of a ConvTranspose2d weight [in, out, k, k]
        sd[key + ".weight"] = torch.randn(shape) * math.sqrt(2.0 / fan_out)
        sd[key + ".bias"] = torch.zeros(cout)

    def block(key, cin, mid, cout, se):
        conv(key + ".conv.0", cin, mid, 3)
        conv(key + ".conv.2", mid, cout, 3)
        if se:
            conv(key + ".seblock.conv1", cout, cout // 8, 1)
            conv(key + ".seblock.conv2", cout // 8, cout, 1)

    block("unet1.conv1", in_channels, 32, 64, False)
    conv("unet1.conv1_down", 64, 64, 2)
    block("unet1.conv2", 64, 128, 64, True)
    conv("unet1.conv2_up", 64, 64, 2, transposed=True)
    conv("unet1.conv3", 64, 64, 3)
    if up:
        conv("unet1.conv_bottom", 64, out_channels, 4, transposed=True)
    else:
        conv("unet1.conv_bottom", 64, out_channels, 3)
    block("unet2.conv1", out_channels, 32, 64, False)
    conv("unet2.conv1_down", 64, 64, 2)
    block("unet2.conv2", 64, 64, 128, True)
    conv("unet2.conv2_down", 128, 128, 2)
    block("unet2.conv3", 128, 256, 128, True)
    conv("unet2.conv3_up", 128, 128, 2, transposed=True)
    block("unet2.conv4", 128, 64, 64, True)
    conv("unet2.conv4_up", 64, 64, 2, transposed=True)
    conv("unet2.conv5", 64, 64, 3)
    conv("unet2.conv_bottom", 64, out_channels, 3)
    return sd


class HipCUNetEngine:
    """One ``nunif_cunet`` handle: CUNet / UpCUNet, or the plain conv stacks vgg_7 / upconv_7 (the C side tells them apart
    by the state-dict keys); ``scale`` / ``offset`` are the model's i2i geometry (output side = T * scale - 2 * offset)."""

    def __init__(self, state_dict, no_clip, device, scale=None, offset=None):
        self.device = torch.device(device)
        if scale is None:
            up = state_dict["unet1.conv_bottom.weight"].shape[2] == 4
            scale, offset = (2, 36) if up else (1, 28)
        self.scale, self.offset = scale, offset
        if self.device.type != "cuda":
            raise RuntimeError("the cunet HIP engine needs a ROCm device (model.to('cuda:N')); no CPU fallback")
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
            _hip.check(_hip.lib().nunif_hip_cunet_create(arr, len(descs), 1 if no_clip else 0, ctypes.byref(handle)))
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().nunif_hip_cunet_destroy(h)
            except Exception:
                pass

    def forward(self, x):
        B, C, T, T2 = x.shape
        assert C == 3 and T == T2
        To = T * self.scale - 2 * self.offset
        z = torch.empty((B, 3, To, To), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_cunet_forward(self.handle, ctypes.c_void_p(x.data_ptr()),
                                                          ctypes.c_void_p(z.data_ptr()), B, T,
                                                          _hip.current_stream_ptr(self.device)))
        return z

    def render(self, x, tile_size, batch_size):
        C, H, W = x.shape
        assert C == 3
        sc = self.scale
        y = torch.empty((3, H * sc, W * sc), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_cunet_render(self.handle, ctypes.c_void_p(x.data_ptr()),
                                                         ctypes.c_void_p(y.data_ptr()), H, W, tile_size, batch_size,
                                                         _hip.current_stream_ptr(self.device)))
        return y


class _CUNetBase(I2IBaseModel):
    _up = False

    def __init__(self, in_channels=3, out_channels=3, no_clip=False):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, no_clip=no_clip),
                         scale=2 if self._up else 1, offset=36 if self._up else 28, in_channels=in_channels)
        if in_channels != 3 or out_channels != 3:
            raise ValueError("the HIP cunet engine supports in_channels = out_channels = 3")
        self.register_tile_size_validator(tile_size_validator)
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self.no_clip = no_clip
        self._weights = _init_weights(in_channels, out_channels, self._up)
        self._engine = None

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
            self._engine = HipCUNetEngine(self._weights, self.no_clip, dev)
        return self._engine

    def forward(self, x):
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        dtype = x.dtype
        return self.engine().forward(x.to(device=self.get_device(), dtype=torch.float32).contiguous()).to(dtype)

    def render_frame(self, x, tile_size, batch_size):
        return self.engine().render(x.to(device=self.get_device(), dtype=torch.float32).contiguous(), tile_size, batch_size)


@register_model
class CUNet(_CUNetBase):
    name = "waifu2x.cunet"
    _up = False


@register_model
class UpCUNet(_CUNetBase):
    name = "waifu2x.upcunet"
    _up = True
