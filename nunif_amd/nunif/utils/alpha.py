"""Bleed RGB into the transparent border before super-resolution, on the HIP engine.  Mirrors ``nunif/utils/alpha.py``
``AlphaBorderPadding`` :32-57 (``ChannelWiseSum`` :5-29 is the 3x3 zero-padded box sum): one
``nunif_hip_alpha_border_padding`` call = all ``offset`` iterations (one fused kernel per iteration: both box sums, the
divide, the masked write and the mask growth)."""
import ctypes

import torch

from ... import _hip


class AlphaBorderPadding(torch.nn.Module):
    def forward(self, rgb, alpha, offset):
        assert rgb.ndim == 3 and alpha.ndim == 3 and rgb.shape[0] == 3 and alpha.shape[0] == 1
        if rgb.device.type != "cuda":
            raise RuntimeError(f"AlphaBorderPadding: tensors must live on a ROCm device (got {rgb.device}); no CPU fallback")
        dtype = rgb.dtype
        r = rgb.to(torch.float32).contiguous()
        a = alpha.to(device=r.device, dtype=torch.float32).contiguous()
        _, h, w = r.shape
        out = torch.empty_like(r)
        work = torch.empty(8 * h * w, dtype=torch.float32, device=r.device)
        with torch.cuda.device(r.device):
            _hip.check(_hip.lib().nunif_hip_alpha_border_padding(ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(a.data_ptr()),
                                                                 ctypes.c_void_p(out.data_ptr()),
                                                                 ctypes.c_void_p(work.data_ptr()), h, w, int(offset),
                                                                 _hip.current_stream_ptr(r.device)))
        return out.to(dtype)
