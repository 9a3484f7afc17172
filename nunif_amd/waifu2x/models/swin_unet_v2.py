"""waifu2x swin_unet_v2 ("winc_unet") on the HIP engine.

Mirrors the model classes of ``waifu2x/models/swin_unet_v2.py`` (reference): ``SwinUNet1xV2`` :361-388, ``SwinUNet2xV2``
:391-430, ``SwinUNet4xV2`` :433-475, ``SwinUNetV2Downscaled`` :492-527 and ``tile_size_validator`` :355-358 — same registry
names and aliases, constructor kwargs, ``i2i_*`` geometry (offset 9 s, blend 4 s) and ``state_dict`` keys (including the
``relative_bias.index`` / ``.delta`` buffers), so a reference ``.pth`` of this family loads unchanged.  The forward pass is
``nunif_hip_swin_unet_v2_forward`` (nunif_amd/csrc/swin_unet_v2.hip); Python holds the fp32 master weights only.

Not carried: the training-time ``tile_mode`` / ``tile_2x2_mode`` split (:309-335, a train-step memory trick — ``set_tile_mode``
raises) and the unregistered experimental ``SwinUNetV2Downscaled.from_4x`` weight surgery.
"""
import copy
import ctypes
from collections import OrderedDict

import torch

from ...nunif.models import register_model_factory, I2IBaseModel, register_model
from ... import _hip
from ...synthetic import window_score_bias_input
from .swin_unet import _FlatWeightsModel, tile_size_validator


def _init_weights(scale_factor, base_dim, lv1_mlp_ratio, lv2_mlp_ratio, lv2_ratio, first_layers, last_layers):
    """Fresh weights in the reference's key layout (SwinUNetV2Base.__init__ :272-312): kaiming convs with zero biases
    (basic_module_init), unit norm weights, nearest-neighbour resampling and a zero ``scale_bias`` (SourceResidual :216-243) —
    a freshly constructed model is the nearest-neighbour upscaler, like the reference's."""
    sd = OrderedDict()
    C, C2 = base_dim, int(base_dim * lv2_ratio)

    def conv(key, cin, cout, k, bias=True):
        w = torch.empty(cout, cin, k, k)
        torch.nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        sd[key + ".weight"] = w
        if bias:
            sd[key + ".bias"] = torch.zeros(cout)

    def linear(key, cin, cout):
        w = torch.empty(cout, cin)
        torch.nn.init.trunc_normal_(w, std=0.02)
        sd[key + ".weight"], sd[key + ".bias"] = w, torch.zeros(cout)

    def wac(p, dim, ws, mlp_ratio, conv_mlp=True):
        linear(p + "mha.mha.qkv_proj", dim, dim * 3)
        linear(p + "mha.mha.head_proj", dim, dim)
        index, delta = window_score_bias_input((ws, ws))
        sd[p + "relative_bias.index"], sd[p + "relative_bias.delta"] = index, delta
        linear(p + "relative_bias.to_bias.0", 2, 2 * ws)
        linear(p + "relative_bias.to_bias.2", 2 * ws, 1)
        sd[p + "norm.weight"] = torch.ones(dim)
        mid = int(dim * mlp_ratio)
        conv(p + "conv_mlp.w1", dim, mid, 1)
        if conv_mlp:
            conv(p + "conv_mlp.w2", mid // 2, dim, 3)
        else:
            conv(p + "conv_mlp.w2", mid, dim, 1)

    P = "unet."
    conv(P + "ir.path1.0", 3, 16, 3)
    conv(P + "ir.path2.1", 12, 64, 1)
    wac(P + "ir.path2.2.", 64, 8, 1)
    wac(P + "ir.path2.3.", 64, 8, 1)
    conv(P + "patch", 32, C, 3)
    for i in range(first_layers):
        wac(f"{P}wac1.blocks.{i}.", C, [8, 6][i], lv1_mlp_ratio)
    conv(P + "down1.conv", C, C2, 2)
    for i in range(4):
        wac(f"{P}wac2.blocks.{i}.", C2, 8, lv2_mlp_ratio)
    conv(P + "up1.proj", C2, C * 4, 1)
    for i in range(last_layers):
        wac(f"{P}wac3.blocks.{i}.", C, 8, lv1_mlp_ratio, conv_mlp=i < last_layers - 1)
    conv(P + "to_residual_image.proj", C, 3 * scale_factor ** 2, 1)
    sd[P + "to_image.scale_bias"] = torch.zeros(1)
    w = torch.zeros(3 * scale_factor ** 2, 3, 3, 3)
    for c in range(3):
        w[c * scale_factor ** 2:(c + 1) * scale_factor ** 2, c, 1, 1] = 1.0          # nearest_neighbor_init :228-242
    sd[P + "to_image.resampling.weight"] = w
    return sd


class HipSwinUNetV2Engine:
    """Owns one ``nunif_swin_unet_v2*`` handle (device weights + workspace) for one device."""

    def __init__(self, state_dict, scale_factor, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the swin_unet_v2 HIP engine needs a ROCm device (model.to('cuda:N')); no CPU fallback")
        self.scale_factor = scale_factor
        keep, descs = [], []
        for name, t in state_dict.items():
            if not t.is_floating_point() or name.endswith("relative_bias.delta"):
                continue        # the score-bias index / offsets are recomputed from the window geometry
            t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            keep.append(t)
            d = _hip.TensorDesc()
            d.name = name.encode()
            d.data = t.data_ptr()
            d.ndim = t.dim()
            for i, s in enumerate(t.shape):
                d.shape[i] = s
            descs.append(d)
        arr = (_hip.TensorDesc * len(descs))(*descs)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_swin_unet_v2_create(arr, len(descs), scale_factor, ctypes.byref(handle)))
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().nunif_hip_swin_unet_v2_destroy(h)
            except Exception:
                pass

    def forward(self, x, clamp=True):
        B, C, T, T2 = x.shape
        assert C == 3 and T == T2
        s = self.scale_factor
        o = (T - 18) * s
        z = torch.empty((B, 3, o, o), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_swin_unet_v2_forward(
                self.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(z.data_ptr()), B, T, 1 if clamp else 0,
                _hip.current_stream_ptr(self.device)))
        return z


class _HipSwinUNetV2Model(_FlatWeightsModel):
    def _setup(self, in_channels, out_channels, base_dim, lv1_mlp_ratio, lv2_mlp_ratio, lv2_ratio, first_layers=2, last_layers=3):
        if in_channels != 3 or out_channels != 3:
            raise ValueError("the HIP swin_unet_v2 engine supports in_channels = out_channels = 3")
        C2 = base_dim * lv2_ratio
        if (base_dim not in (64, 96, 128) or C2 != int(C2) or int(C2) not in (128, 192, 256) or lv1_mlp_ratio != 2
                or lv2_mlp_ratio not in (1, 2) or first_layers not in (1, 2) or last_layers < 1):
            raise ValueError("the HIP swin_unet_v2 engine carries base_dim 64 / 96 / 128 with a level-2 width of 128 / 192 / 256, "
                             "lv1_mlp_ratio 2, lv2_mlp_ratio 1 / 2 and 1-2 first layers (the registered 1x / 2x / 4x models)")
        self._setup_weights(_init_weights(self.unet_scale_factor, base_dim, lv1_mlp_ratio, lv2_mlp_ratio, lv2_ratio,
                                          first_layers, last_layers))

    def _make_engine(self, device):
        return HipSwinUNetV2Engine(self._weights, self.unet_scale_factor, device)

    def set_tile_mode(self):
        raise NotImplementedError("tile_mode is a training-time split (swin_unet_v2.py:309-335); the HIP engine is inference-only")

    set_tile_2x2_mode = set_tile_mode


@register_model
class SwinUNet1xV2(_HipSwinUNetV2Model):
    name = "waifu2x.swin_unet_v2_1x"
    name_alias = ("waifu2x.winc_unet_1x", "waifu2x.swin_unet_1x_v2")
    unet_scale_factor = 1

    def __init__(self, in_channels=3, out_channels=3, base_dim=64, lv1_mlp_ratio=2, lv2_mlp_ratio=2, lv2_ratio=2,
                 first_layers=2, last_layers=3, **kwargs):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, base_dim=base_dim, lv1_mlp_ratio=lv1_mlp_ratio,
                              lv2_mlp_ratio=lv2_mlp_ratio, lv2_ratio=lv2_ratio, first_layers=first_layers,
                              last_layers=last_layers, **kwargs),
                         scale=1, offset=9, in_channels=in_channels, blend_size=4)
        self._setup(in_channels, out_channels, base_dim, lv1_mlp_ratio, lv2_mlp_ratio, lv2_ratio, first_layers, last_layers)


@register_model
class SwinUNet2xV2(_HipSwinUNetV2Model):
    name = "waifu2x.swin_unet_v2_2x"
    name_alias = ("waifu2x.winc_unet_2x",)
    unet_scale_factor = 2

    def __init__(self, in_channels=3, out_channels=3, base_dim=96, lv1_mlp_ratio=2, lv2_mlp_ratio=2, lv2_ratio=2, **kwargs):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, base_dim=base_dim, lv1_mlp_ratio=lv1_mlp_ratio,
                              lv2_mlp_ratio=lv2_mlp_ratio, lv2_ratio=lv2_ratio, **kwargs),
                         scale=2, offset=18, in_channels=in_channels, blend_size=8)
        self._setup(in_channels, out_channels, base_dim, lv1_mlp_ratio, lv2_mlp_ratio, lv2_ratio)


@register_model
class SwinUNet4xV2(_HipSwinUNetV2Model):
    name = "waifu2x.swin_unet_v2_4x"
    name_alias = ("waifu2x.winc_unet_4x",)
    unet_scale_factor = 4

    def __init__(self, in_channels=3, out_channels=3, base_dim=128, lv1_mlp_ratio=2, lv2_mlp_ratio=2, lv2_ratio=2, **kwargs):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, base_dim=base_dim, lv1_mlp_ratio=lv1_mlp_ratio,
                              lv2_mlp_ratio=lv2_mlp_ratio, lv2_ratio=lv2_ratio, **kwargs),
                         scale=4, offset=36, in_channels=in_channels, blend_size=16)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self._setup(in_channels, out_channels, base_dim, lv1_mlp_ratio, lv2_mlp_ratio, lv2_ratio)

    def to_2x(self, shared=True):
        unet = self if shared else copy.deepcopy(self)
        return SwinUNetV2Downscaled(unet, downscale_factor=2, in_channels=self.i2i_in_channels, out_channels=self.out_channels)

    def to_1x(self, shared=True):
        unet = self if shared else copy.deepcopy(self)
        return SwinUNetV2Downscaled(unet, downscale_factor=4, in_channels=self.i2i_in_channels, out_channels=self.out_channels)


def _swin_unet_v2_1xs(**kwargs):
    """``waifu2x.swin_unet_v2_1xs`` (reference :528-530: base_dim 32, one first / last layer, mlp ratios 1) — an experimental
    geometry without released weights that the HIP engine does not carry (its kernels are instantiated for base_dim 64 / 96 /
    128 with lv1_mlp_ratio 2).  Registered so that the name is KNOWN and fails with the reason instead of "Unknown model name"."""
    raise NotImplementedError("waifu2x.swin_unet_v2_1xs (base_dim 32, lv1_mlp_ratio 1) is not on the HIP engine: it carries the "
                              "registered 1x / 2x / 4x geometries (base_dim 64 / 96 / 128, lv1_mlp_ratio 2)")


_swin_unet_v2_1xs._nunif_amd_unsupported = True          # nunif_amd.install() leaves the reference's own factory for this name in place
register_model_factory("waifu2x.swin_unet_v2_1xs", _swin_unet_v2_1xs)


@register_model
class SwinUNetV2Downscaled(I2IBaseModel):
    """The 4x net followed by bicubic-antialias /f, clamped before and after (reference :492-527, eval branch)."""
    name = "waifu2x.swin_unet_v2_downscaled"

    def __init__(self, unet, downscale_factor, in_channels=3, out_channels=3):
        assert downscale_factor in {2, 4}
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, downscale_factor=downscale_factor),
                         scale=4 // downscale_factor, offset={2: 18, 4: 9}[downscale_factor], in_channels=in_channels,
                         blend_size=4 * downscale_factor)
        self.register_tile_size_validator(tile_size_validator)
        self.net4x = unet
        self.downscale_factor = downscale_factor
        self.mode, self.antialias = "bicubic", True

    def get_device(self):
        return self.net4x.get_device()

    def state_dict(self, *args, **kwargs):
        return self.net4x.state_dict()

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        return self.net4x.load_state_dict(state_dict, strict=strict)

    def forward(self, x):
        from ...iw3 import _ops
        z = self.net4x(x)           # already clamped to [0,1]
        f = self.downscale_factor
        return _ops.resize_aa(z, (z.shape[-2] // f, z.shape[-1] // f), mode="bicubic", align_corners=False,
                              clamp01=True).to(x.dtype)
