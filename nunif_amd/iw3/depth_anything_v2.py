"""Depth-Anything (V1 / V2 / V2-metric, ViT-S / B / L) backbone on the HIP engine (``nunif_hip_depth_anything_*``, nunif_amd/csrc/depth_anything.hip).

The reference obtains this network with ``torch.hub.load("nagadomi/Depth-Anything_iw3", "DepthAnything", encoder=...)``
(``iw3/depth_anything_model.py:200-230``) and calls it as ``model(x)`` on the ImageNet-normalised, /14-aligned batch from
``batch_preprocess`` (``_forward`` :113-119).  Neither that repository nor its weights are reachable offline, so this
class follows the PUBLISHED architecture and checkpoint key layout (``pretrained.*`` DINOv2 ViT-S/14, ``depth_head.*``
DPT) — see ``oracle/depth_anything_v2.py``.  Since round 3 that restatement is pinned against HuggingFace ``transformers``
(``tests/test_depth_anything_vs_hf.py``, <= 5e-5 for ViT-S / B / L, V1 taps, the metric head) and this engine against
HuggingFace-produced fixtures (``tests/golden/depth_anything_hf.npz``); what nobody can check offline is the hub FORK itself —
first of all its position-embedding resize (upstream ``scale_factor=(g + 0.1) / 37``, restated from memory).

The geometry (embed 384 / 768 / 1024, 12 / 24 blocks, DPT widths) is read from the checkpoint; ``taps`` (the four encoder blocks
that feed the head: V2 default, V1 = the last four) and ``max_depth`` (> 0: the V2 metric head, Sigmoid x max_depth) are what
the hub entry points ``DepthAnything`` / ``DepthAnythingMetricDepthV2`` decide from the model name.

``HipDepthAnythingV2(state_dict)(x[B,3,h,w]) -> [B,h,w]`` is a drop-in ``backbone`` for
``nunif_amd.iw3.base_depth_model.CallableDepthModel`` (pre/post-processing, TTA flip, edge dilation, DepthAA, EMA
normalisation stay in the shared pipeline).
"""
import ctypes
import math

import torch
import torch.nn.functional as F

from .. import _hip

PATCH = 14


def interpolate_pos_embed(pos_embed, gh, gw):
    """DINOv2 ``interpolate_pos_encoding`` (bicubic, +0.1 offset, no antialias) — weight preparation, once per grid."""
    EMBED = pos_embed.shape[-1]
    n = pos_embed.shape[1] - 1
    s = int(math.sqrt(n))
    if gh == s and gw == s:
        return pos_embed
    cls, patch = pos_embed[:, :1], pos_embed[:, 1:]
    patch = patch.reshape(1, s, s, EMBED).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, scale_factor=((gh + 0.1) / s, (gw + 0.1) / s), mode="bicubic", antialias=False)
    assert patch.shape[-2:] == (gh, gw)
    return torch.cat([cls, patch.permute(0, 2, 3, 1).reshape(1, gh * gw, EMBED)], dim=1)


class HipDepthAnythingV2:
    metric_depth = False

    def __init__(self, state_dict, device="cuda:0", taps=None, max_depth=0.0):
        self.device = torch.device(device)
        self.taps = None if taps is None else tuple(int(t) for t in taps)
        self.max_depth = float(max_depth or 0.0)
        self.metric_depth = self.max_depth > 0
        if self.device.type != "cuda":
            raise RuntimeError("the Depth-Anything HIP engine needs a ROCm device; there is no CPU fallback")
        self._state_dict = state_dict             # host tensors; kept so that replica() can build the same engine elsewhere
        self._pos_embed = state_dict["pretrained.pos_embed"].detach().float().cpu()
        self._pos_cache = {}
        keep, descs = [], []
        for name, t in state_dict.items():
            if not torch.is_floating_point(t) or name == "pretrained.pos_embed":
                continue
            t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            keep.append(t)
            d = _hip.TensorDesc()
            d.name, d.data, d.ndim = name.encode(), t.data_ptr(), min(t.dim(), 4)
            for i, s in enumerate(t.shape[:4]):
                d.shape[i] = s
            descs.append(d)
        arr = (_hip.TensorDesc * len(descs))(*descs)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            taps_arr = None if self.taps is None else (ctypes.c_int32 * 4)(*self.taps)
            _hip.check(_hip.lib().nunif_hip_depth_anything_create_ex(arr, len(descs), taps_arr, ctypes.c_float(self.max_depth),
                                                                    ctypes.byref(handle)))
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().nunif_hip_depth_anything_destroy(h)
            except Exception:
                pass

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def replica(self, device):
        """The same network on another device (its own engine handle and workspace): what
        ``nunif.models.data_parallel.DeviceSwitchInference`` keeps one of per listed GPU."""
        return self if torch.device(device) == self.device else type(self)(self._state_dict, device, self.taps, self.max_depth)

    def _pos(self, gh, gw):
        key = (gh, gw)
        if key not in self._pos_cache:
            self._pos_cache[key] = interpolate_pos_embed(self._pos_embed, gh, gw)[0].contiguous().to(self.device)
        return self._pos_cache[key]

    @torch.inference_mode()
    def __call__(self, x):
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, C, h, w = x.shape
        if C != 3 or h % PATCH or w % PATCH:
            raise ValueError(f"expected [B,3,h,w] with h, w multiples of {PATCH}, got {tuple(x.shape)}")
        pos = self._pos(h // PATCH, w // PATCH)
        out = torch.empty((B, h, w), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().nunif_hip_depth_anything_forward(self.handle, ctypes.c_void_p(x.data_ptr()),
                                                                   ctypes.c_void_p(pos.data_ptr()),
                                                                   ctypes.c_void_p(out.data_ptr()), B, h, w,
                                                                   _hip.current_stream_ptr(self.device)))
        return out
