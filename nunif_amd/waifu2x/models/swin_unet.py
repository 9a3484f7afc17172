"""waifu2x swin_unet on the HIP engine.

Mirrors the model classes of ``waifu2x/models/swin_unet.py`` (reference): ``SwinUNet`` :208-226, ``SwinUNet2x``
:229-249, ``SwinUNet4x`` :261-306, ``SwinUNetDownscaled`` :339-387 and ``tile_size_validator`` :202-205 — same
registry names, constructor kwargs, ``i2i_*`` geometry and ``state_dict`` keys, so reference ``.pth`` files load
unchanged.  The forward pass is ``nunif_hip_swin_unet_forward`` (nunif_amd/csrc/swin_unet.cpp); Python only holds
the fp32 master weights and hands their pointers to the C ABI.
"""
import ctypes
import math
from collections import OrderedDict

import copy

import torch

from ...nunif.models import I2IBaseModel, register_model, register_model_factory
from ... import _hip

WINDOW = 6


def tile_size_validator(size):
    return size > 16 and (size - 16) % 12 == 0 and (size - 16) % 16 == 0


def _relative_position_index():
    ys, xs = torch.meshgrid(torch.arange(WINDOW), torch.arange(WINDOW), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    return ((ys[:, None] - ys[None, :] + WINDOW - 1) * (2 * WINDOW - 1) + xs[:, None] - xs[None, :] + WINDOW - 1).reshape(-1)


def _init_weights(scale_factor, base_dim, in_channels, out_channels, layer_norm=False):
    """Fresh weights in the reference's key layout (SwinUNetBase.__init__ :119-178; torchvision block init)."""
    sd = OrderedDict()
    C = base_dim

    def conv(key, cin, cout, k):
        w = torch.empty(cout, cin, k, k)
        torch.nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        sd[key + ".weight"], sd[key + ".bias"] = w, torch.zeros(cout)

    def linear(key, cin, cout, bias_std=0.0):
        w = torch.empty(cout, cin)
        torch.nn.init.xavier_uniform_(w)
        sd[key + ".weight"] = w
        sd[key + ".bias"] = torch.randn(cout) * bias_std if bias_std else torch.zeros(cout)

    def stage(key, dim, layers):
        for i in range(layers):
            p = f"{key}.block.{i}."
            for name, cout in (("attn.qkv", dim * 3), ("attn.proj", dim)):
                w = torch.empty(cout, dim)
                torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                sd[p + name + ".weight"] = w
                sd[p + name + ".bias"] = (torch.rand(cout) * 2 - 1) / math.sqrt(dim)
            sd[p + "attn.relative_position_bias_table"] = torch.nn.init.trunc_normal_(
                torch.empty((2 * WINDOW - 1) ** 2, C // 16), std=0.02)
            sd[p + "attn.relative_position_index"] = _relative_position_index()
            linear(p + "mlp.0", dim, dim * 2, 1e-6)
            linear(p + "mlp.3", dim * 2, dim, 1e-6)
            if layer_norm:      # LayerNormNoBias: weight only (nunif/modules/norm.py:17-22)
                sd[p + "norm1.weight"] = torch.ones(dim)
                sd[p + "norm2.weight"] = torch.ones(dim)

    P = "unet."
    conv(P + "patch.0", in_channels, C // 2, 3)
    conv(P + "patch.2", C // 2, C, 3)
    stage(P + "swin1", C, 2)
    conv(P + "down1.conv", C, C * 2, 2)
    stage(P + "swin2", C * 2, 2)
    conv(P + "down2.conv", C * 2, C * 2, 2)
    stage(P + "swin3", C * 2, 6)
    linear(P + "up2.proj", C * 2, C * 2 * 4)
    if scale_factor in (4, 8):
        linear(P + "proj2", C, C * 2)
    stage(P + "swin4", C * 2, 2)
    top = C if scale_factor in (1, 2) else C * 2
    linear(P + "up1.proj", C * 2, top * 4)
    stage(P + "swin5", top, 2)
    if scale_factor == 8:            # ToImage :96-101: Linear -> LeakyReLU(0.2) -> Linear
        linear(P + "to_image.proj.0", top, out_channels * 64)
        linear(P + "to_image.proj.2", out_channels * 64, out_channels * 64)
    else:
        linear(P + "to_image.proj", top, out_channels * scale_factor ** 2)
    return sd


class HipSwinUNetEngine:
    """Owns one ``nunif_swin_unet*`` handle (device weights + workspace) for one device."""

    def __init__(self, state_dict, scale_factor, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the swin_unet HIP engine needs a ROCm device (model.to('cuda:N')); no CPU fallback")
        self.scale_factor = scale_factor
        keep = []
        descs = []
        for name, t in state_dict.items():
            if not t.is_floating_point():
                continue   # relative_position_index is recomputed from the window geometry
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
            _hip.check(_hip.lib().nunif_hip_swin_unet_create(arr, len(descs), scale_factor, ctypes.byref(handle)))
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().nunif_hip_swin_unet_destroy(h)
            except Exception:
                pass

    def forward(self, x):
        B, C, T, T2 = x.shape
        assert C == 3 and T == T2
        s = self.scale_factor
        z = torch.empty((B, 3, (T - 16) * s, (T - 16) * s), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_swin_unet_forward(
                self.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(z.data_ptr()), B, T,
                _hip.current_stream_ptr(self.device)))
        return z

    def render(self, x, tile_size, batch_size):
        C, H, W = x.shape
        assert C == 3
        s = self.scale_factor
        y = torch.empty((3, H * s, W * s), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_swin_unet_render(
                self.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), H, W, tile_size,
                batch_size, _hip.current_stream_ptr(self.device)))
        return y


class SwinRowEngine:
    """The tile-row interface ``nunif_amd.parallel.render_rows_sharded`` drives (one huge image over several GPUs): the
    whole-frame render of ``engine`` cut into partial renders of tile rows, band export / import and a row-band stitch."""

    def __init__(self, engine, H, W, tile_size, batch_size):
        self.e, self.H, self.W, self.tile_size, self.batch_size = engine, H, W, tile_size, batch_size
        s = engine.scale_factor
        self.grid = _hip.tile_grid(H, W, s, 8 * s, tile_size, 4 * s)
        self.h_blocks, self.w_blocks = self.grid.h_blocks, self.grid.w_blocks
        self.out_tile_size, self.output_tile_step = self.grid.out_tile_size, self.grid.output_tile_step
        self.y_h, self.y_w = self.grid.y_h, self.grid.y_w
        self.device = engine.device

    def _call(self, fn, *args):
        with torch.cuda.device(self.device):
            _hip.check(fn(*args, _hip.current_stream_ptr(self.device)))

    def render_tile_rows(self, x, r0, r1):
        self._call(_hip.lib().nunif_hip_swin_unet_render_tile_rows, self.e.handle, ctypes.c_void_p(x.data_ptr()), self.H, self.W,
                   self.tile_size, self.batch_size, r0, r1)

    def export_band(self, tile_row, row0, n_rows):
        band = torch.empty((self.w_blocks, 3, n_rows, self.out_tile_size), dtype=torch.float32, device=self.device)
        self._call(_hip.lib().nunif_hip_swin_unet_tile_row_band, self.e.handle, self.H, self.W, self.tile_size, tile_row, row0,
                   n_rows, ctypes.c_void_p(band.data_ptr()), 0)
        return band

    def import_band(self, tile_row, row0, band):
        band = band.to(device=self.device, dtype=torch.float32).contiguous()
        self._call(_hip.lib().nunif_hip_swin_unet_tile_row_band, self.e.handle, self.H, self.W, self.tile_size, tile_row, row0,
                   band.shape[2], ctypes.c_void_p(band.data_ptr()), 1)

    def stitch_rows(self, y0, y1):
        out = torch.empty((3, y1 - y0, self.y_w), dtype=torch.float32, device=self.device)
        if y1 > y0:
            self._call(_hip.lib().nunif_hip_swin_unet_stitch_rows, self.e.handle, ctypes.c_void_p(out.data_ptr()), self.H, self.W,
                       self.tile_size, y0, y1)
        return out


class _FlatWeightsModel(I2IBaseModel):
    """Common machinery: flat fp32 master weights under the reference's keys + a lazily built HIP engine
    (``_make_engine(device)``, one handle per device)."""
    unet_scale_factor = 1

    def _setup_weights(self, weights):
        self.register_tile_size_validator(tile_size_validator)
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self._weights = weights
        self._engine = None

    def _make_engine(self, device):
        raise NotImplementedError

    # -- nn.Module surface ------------------------------------------------------------------------------------
    def get_device(self):
        return self._device_probe.device

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._weights.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        missing = [k for k in self._weights if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._weights]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: "
                               f"missing {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                               f"unexpected {unexpected[:4]}{'...' if len(unexpected) > 4 else ''}")
        for k in self._weights:
            if k in state_dict:
                v = state_dict[k].detach().to("cpu")
                if v.shape != self._weights[k].shape:
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._weights[k].shape)}")
                self._weights[k] = v.to(self._weights[k].dtype).clone()
        self._engine = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def parameters(self, recurse=True):
        return iter(v for v in self._weights.values() if v.is_floating_point())

    def half(self):      # storage precision is the engine's business (fp16 maps, fp32 accumulate)
        return self

    def float(self):
        return self

    def engine(self):
        dev = self.get_device()
        if self._engine is None or self._engine.device != dev:
            self._engine = self._make_engine(dev)
        return self._engine

    def _prepare(self, x):
        return x.to(device=self.get_device(), dtype=torch.float32).contiguous()

    def forward(self, x):
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        dtype = x.dtype
        return self.engine().forward(self._prepare(x)).to(dtype)


class _HipSwinUNetModel(_FlatWeightsModel):
    def _setup(self, in_channels, out_channels, base_dim=96, layer_norm=False):
        if in_channels != 3 or out_channels != 3:
            raise ValueError("the HIP swin_unet engine supports in_channels = out_channels = 3")
        if base_dim not in (96, 192):
            raise ValueError("the HIP swin_unet engine supports base_dim 96 and 192 (swin_unet_4xl)")
        self._setup_weights(_init_weights(self.unet_scale_factor, base_dim, in_channels, out_channels, layer_norm))

    def _make_engine(self, device):
        return HipSwinUNetEngine(self._weights, self.unet_scale_factor, device)

    def render_frame(self, x, tile_size, batch_size):
        """Whole-frame tiled render in one C call (used by SeamBlending.tiled_render)."""
        return self.engine().render(self._prepare(x), tile_size, batch_size)

    def row_engine(self, x, tile_size=None, batch_size=None):
        """-> (prepared frame, SwinRowEngine): the tile-row form of ``render_frame`` for ``parallel.render_rows_sharded``."""
        x = self._prepare(x)
        return x, SwinRowEngine(self.engine(), x.shape[1], x.shape[2], self.find_valid_tile_size(tile_size),
                                batch_size or self.i2i_default_batch_size)


@register_model
class SwinUNet(_HipSwinUNetModel):
    name = "waifu2x.swin_unet_1x"
    unet_scale_factor = 1

    def __init__(self, in_channels=3, out_channels=3):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels),
                         scale=1, offset=8, in_channels=in_channels, blend_size=4)
        self._setup(in_channels, out_channels)


@register_model
class SwinUNet2x(_HipSwinUNetModel):
    name = "waifu2x.swin_unet_2x"
    unet_scale_factor = 2

    def __init__(self, in_channels=3, out_channels=3, base_dim=96, layer_norm=False):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, base_dim=base_dim,
                              layer_norm=layer_norm),
                         scale=2, offset=16, in_channels=in_channels, blend_size=8)
        self._setup(in_channels, out_channels, base_dim, layer_norm)


@register_model
class SwinUNet4x(_HipSwinUNetModel):
    name = "waifu2x.swin_unet_4x"
    unet_scale_factor = 4

    def __init__(self, in_channels=3, out_channels=3, pre_antialias=False, base_dim=96, layer_norm=False):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, pre_antialias=pre_antialias,
                              base_dim=base_dim, layer_norm=layer_norm),
                         scale=4, offset=32, in_channels=in_channels, blend_size=16)
        self.out_channels = out_channels
        self.pre_antialias = pre_antialias
        self.antialias = True
        self._setup(in_channels, out_channels, base_dim, layer_norm)

    def __getattribute__(self, name):
        # with pre_antialias every TILE is resized before the net (reference :281-282), so the fused whole-frame
        # render (tile gather inside the first conv) does not apply: hide it and let tiled_render loop over tiles
        if name == "render_frame" and object.__getattribute__(self, "__dict__").get("pre_antialias", False):
            raise AttributeError(name)
        return super().__getattribute__(name)

    def forward(self, x):
        if self.pre_antialias:
            # resize_antialias (reference :252-258): bicubic x2 up, then bicubic /2 down, both with antialias
            from ...iw3 import _ops
            h, w = x.shape[-2:]
            xf = self._prepare(x)
            xf = _ops.resize_aa(xf, (h * 2, w * 2), mode="bicubic", align_corners=False)
            x = _ops.resize_aa(xf, (h, w), mode="bicubic", align_corners=False).to(x.dtype)
        return super().forward(x)

    def to_2x(self, shared=True):
        # reference :283-289: shared=False gives the downscaled model its own copy of the 4x net
        unet = self if shared else copy.deepcopy(self)
        return SwinUNetDownscaled(in_channels=self.i2i_in_channels, out_channels=self.out_channels,
                                  downscale_factor=2, unet=unet)

    def to_1x(self, shared=True):
        unet = self if shared else copy.deepcopy(self)
        return SwinUNetDownscaled(in_channels=self.i2i_in_channels, out_channels=self.out_channels,
                                  downscale_factor=4, unet=unet)


def swin_unet_4xl(**kwargs):
    """Reference :390-394: the 4x net at base_dim 192 (12 heads of 16 / 32) with LayerNormNoBias in every block.  Runs on the
    engine's generic block path (gemm_kernel Linears + window_attn_kernel + layernorm_nobias_kernel)."""
    return SwinUNet4x(base_dim=192, layer_norm=True, **kwargs)


register_model_factory("waifu2x.swin_unet_4xl", swin_unet_4xl)


@register_model
class SwinUNet8x(_HipSwinUNetModel):
    """Reference :303-321.  The registered geometry (scale 4, offset 64, blend 32) is the reference's; the net itself
    returns 8 x (T - 16) pixels per side, so only the per-tile ``forward`` is meaningful (as in the reference)."""
    name = "waifu2x.swin_unet_8x"
    unet_scale_factor = 8

    def __init__(self, in_channels=3, out_channels=3):
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels),
                         scale=4, offset=64, in_channels=in_channels, blend_size=32)
        self._setup(in_channels, out_channels)

    def __getattribute__(self, name):
        if name == "render_frame":             # no consistent tile grid exists for this net
            raise AttributeError(name)
        return super().__getattribute__(name)


@register_model
class SwinUNetDownscaled(I2IBaseModel):
    """4x net followed by bicubic-antialias /f and clamp (reference :339-379)."""
    name = "waifu2x.swin_unet_downscaled"

    def __init__(self, in_channels=3, out_channels=3, downscale_factor=2, unet=None, pre_antialias=False):
        assert downscale_factor in {2, 4}
        super().__init__(dict(in_channels=in_channels, out_channels=out_channels, downscale_factor=downscale_factor),
                         scale=4 // downscale_factor, offset=32 // downscale_factor, in_channels=in_channels,
                         blend_size=4 * downscale_factor)
        self.register_tile_size_validator(tile_size_validator)
        self.net4x = unet if unet is not None else SwinUNet4x(in_channels, out_channels, pre_antialias=pre_antialias)
        self.downscale_factor = downscale_factor

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
        # bicubic (a=-0.5) antialiased /f + clamp, fused in nunif_hip_resize_aa (reference :366-379)
        return _ops.resize_aa(z, (z.shape[-2] // f, z.shape[-1] // f), mode="bicubic", align_corners=False,
                              clamp01=True).to(x.dtype)
