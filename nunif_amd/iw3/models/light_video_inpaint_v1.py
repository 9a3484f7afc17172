"""``inpaint.light_video_inpaint_v1`` on the HIP engine.

Mirrors ``iw3/models/light_video_inpaint_v1.py`` ``LightVideoInpaintV1`` :92-229 (base_dim 96, lv2_mlp_ratio 1): registry
name + alias, i2i geometry (scale 1, offset 16, blend 8), the ``state_dict`` key layout and ``infer`` :140-164 — the batch is
padded to ``SEQ_LEN`` = 12 frames by repeating the first / last frame, pre-processed and run through the net whose level-2
stack alternates 8x8 window gMLPs with temporal gMLPs over the 12 frames of each pixel.  One C call per 12 frames
(``nunif_hip_light_inpaint_infer``, nunif_amd/csrc/light_inpaint.hip).  ``LightVideoInpaintV1Medium`` (base_dim 128) and
``LightVideoInpaintV1Large`` (base_dim 192), both with lv2_mlp_ratio 2 (:230-246), run on the same engine.
"""
import math
from collections import OrderedDict

import torch

from ...nunif.models import I2IBaseModel, register_model
from .light_inpaint_v1 import HipLightInpaintEngine

SEQ_LEN = 12
OFFSET = 16


def _init_weights(base_dim=96, lv2_mlp_ratio=1):
    sd = OrderedDict()
    C, r2 = base_dim, lv2_mlp_ratio

    def lin(key, *shape):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        sd[key + ".weight"] = torch.randn(shape) * math.sqrt(1.0 / fan_in)
        sd[key + ".bias"] = torch.zeros(shape[0])

    def block(p, C, N, ratio):
        V = C * ratio
        lin(p + "gmlp.gmlp.proj_in", 2 * V, C)
        sd[p + "gmlp.gmlp.proj_spatial.weight"] = (torch.rand(N, N, 1) * 2 - 1) * (1e-3 / C)
        sd[p + "gmlp.gmlp.proj_spatial.bias"] = torch.ones(N)
        lin(p + "gmlp.gmlp.proj_out", C, V)
        sd[p + "norm1.weight"] = torch.ones(C)
        sd[p + "norm2.weight"] = torch.ones(V)
        lin(p + "glu_conv.w1", C, C, 1, 1)
        lin(p + "glu_conv.w2", C, C // 2, 3, 3)

    sd["mask_bias"] = torch.randn(1, C, 1, 1) * 0.01
    lin("patch", C, 3, 4, 4)
    block("enc1.", C, 256, 2)
    lin("down", 2 * C, C, 2, 2)
    for i, (n, r) in enumerate(((64, r2), (SEQ_LEN, 2), (64, r2), (SEQ_LEN, 2), (64, r2))):
        block(f"enc2.{i}.", 2 * C, n, r)
    lin("up", 4 * C, 2 * C, 1, 1)
    block("dec1.", C, 256, 2)
    lin("to_image", 48, C, 1, 1)
    return sd


@register_model
class LightVideoInpaintV1(I2IBaseModel):
    name = "inpaint.light_video_inpaint_v1"
    name_alias = ("inpaint.light_video_inpaint_v1_small",)

    def __init__(self, base_dim=96, lv2_mlp_ratio=1):
        super().__init__(dict(base_dim=base_dim, lv2_mlp_ratio=lv2_mlp_ratio), scale=1, offset=OFFSET, in_channels=3,
                         blend_size=8)
        if (base_dim, lv2_mlp_ratio) not in ((96, 1), (128, 2), (192, 2)):
            raise ValueError("the HIP engine implements the registered variants: (base_dim, lv2_mlp_ratio) = (96, 1) small, "
                             "(128, 2) medium, (192, 2) large")
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)
        self.sequence_offset, self.downscaling_factor, self.mod = 0, 4, 16
        self._weights = _init_weights(base_dim, lv2_mlp_ratio)
        self._engine = None

    def get_device(self):
        return self._device_probe.device

    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.clone()) for k, v in self._weights.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        missing = [k for k in self._weights if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._weights]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for LightVideoInpaintV1: missing {missing[:4]}, "
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
        """x [N,3,H,W] (N <= 12 consecutive frames), mask [N,1,H,W] bool -> inpainted [N,3,H,W]  (reference :140-164).
        ``mirror_x`` (not in the reference): ``flip(infer(flip(x), mask))`` with ``mask`` given in the flipped frame, no flip passes."""
        if self.training:
            raise RuntimeError("the HIP engine is inference-only; call .eval()")
        dev = self.get_device()
        n = x.shape[0]
        if n > SEQ_LEN:
            raise AssertionError(f"at most {SEQ_LEN} frames per call (got {n})")     # _forward asserts x.shape[0] == SEQ_LEN
        pad_b1 = pad_b2 = 0
        if n % SEQ_LEN != 0:
            pad_b = SEQ_LEN - n % SEQ_LEN
            pad_b1, pad_b2 = pad_b // 2, pad_b - pad_b // 2
            x = torch.cat([x[0:1]] * pad_b1 + [x] + [x[-1:]] * pad_b2, dim=0)
            mask = torch.cat([mask[0:1]] * pad_b1 + [mask] + [mask[-1:]] * pad_b2, dim=0)
        W = x.shape[-1]

        def n_iter(k):
            if k <= 0:
                return 0
            return max(round(W / base_width * k), 1) if base_width is not None else k
        m = mask.to(device=dev)
        # the engine takes a byte per pixel, hole = non-zero: a bool mask IS that (zero-copy view), a uint8 mask passes as it is
        if m.dtype == torch.bool:
            m = m.contiguous().view(torch.uint8)
        elif m.dtype != torch.uint8:
            m = (m > 0).to(torch.uint8)
        m = m.contiguous()
        dtype = x.dtype
        out = self.engine().infer(x.to(device=dev, dtype=torch.float32).contiguous(), m, closing,
                                  n_iter(inner_dilation), n_iter(outer_dilation), mirror_x=mirror_x).to(dtype)
        return out[pad_b1:out.shape[0] - pad_b2]

    def forward(self, x, mask, skip_i2i_offset=False, micro_batch_size=SEQ_LEN):
        raise NotImplementedError("the HIP engine implements LightVideoInpaintV1.infer; the training-style forward is not provided")


@register_model
class LightVideoInpaintV1Medium(LightVideoInpaintV1):
    name = "inpaint.light_video_inpaint_v1_medium"
    name_alias = ()

    def __init__(self, base_dim=128, lv2_mlp_ratio=2):
        super().__init__(base_dim=base_dim, lv2_mlp_ratio=lv2_mlp_ratio)


@register_model
class LightVideoInpaintV1Large(LightVideoInpaintV1):
    name = "inpaint.light_video_inpaint_v1_large"
    name_alias = ()

    def __init__(self, base_dim=192, lv2_mlp_ratio=2):
        super().__init__(base_dim=base_dim, lv2_mlp_ratio=lv2_mlp_ratio)
