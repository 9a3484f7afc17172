"""Tile -> model -> stitch loop on the HIP engine.

Mirrors ``nunif/utils/seam_blending.py`` (reference): same class, static methods and ``tiled_render`` signature
(:48-106).  Differences in *mechanism*, not results:

* ``create_config`` (:109-143) is the C-ABI integer routine ``nunif_hip_tile_grid_init`` (bit-exact).
* the reference keeps frame-sized ``pixels`` / ``weights`` accumulators and updates them per tile (:156-174); here
  all tile outputs of the frame stay resident and one HIP kernel replays the same running-mean recurrence per
  output pixel (``nunif_hip_stitch_tiles``) — see nunif_amd/csrc/stitch.hip.
* tile slicing + replicate padding (:82,:90) is ``nunif_hip_gather_tiles`` (or is fused into the model's first
  conv when the model offers ``render_frame``).
"""
import ctypes

import torch

from .. import device as _device
from ..models.utils import get_model_device
from ... import _hip


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class SeamBlending(torch.nn.Module):
    def __init__(self, x_shape, scale, offset, tile_size, blend_size):
        super().__init__()
        C, H, W = x_shape
        self.channels = C
        self.grid = _hip.tile_grid(H, W, scale, offset, tile_size, blend_size)
        g = self.grid
        self.output_tile_step = g.output_tile_step
        self.input_tile_step = g.input_tile_step
        self.h_blocks, self.w_blocks = g.h_blocks, g.w_blocks
        self.y_h, self.y_w = g.y_h, g.y_w
        self.pad = (g.pad_l, g.pad_r, g.pad_t, g.pad_b)
        self.blend_size = blend_size
        self.tile_out = None   # [h_blocks*w_blocks, C, To, To] fp32, allocated on first use

    def _store(self, device):
        if self.tile_out is None or self.tile_out.device != device:
            g = self.grid
            self.tile_out = torch.zeros((g.h_blocks * g.w_blocks, self.channels, g.out_tile_size, g.out_tile_size),
                                        dtype=torch.float32, device=device)
        return self.tile_out

    def forward(self, x, i, j):
        """Deposit the model output of tile (i, j).  (the reference blends here: ``update`` :156-174)"""
        store = self._store(x.device)
        store[i * self.w_blocks + j].copy_(x)
        return x

    def get_output(self):
        assert self.tile_out is not None
        store = self.tile_out
        y = torch.empty((self.channels, self.y_h, self.y_w), dtype=torch.float32, device=store.device)
        _hip.check(_hip.lib().nunif_hip_stitch_tiles(_ptr(store), _ptr(y), ctypes.byref(self.grid), self.channels,
                                                     _hip.current_stream_ptr(store.device)))
        return y

    def clear(self):
        if self.tile_out is not None:
            self.tile_out.zero_()

    def gather(self, x, tile_begin, n_tiles, out, channels=None):
        """out[k] = replicate-padded x[:, i:i+T, j:j+T] for the row-major tile range (reference :82,:90).  ``channels`` is
        the INPUT channel count (x.shape[0]); ``self.channels`` is the stitch store's OUTPUT count, which differs when a
        config_callback changes the channel count (reference :64-72 slices x with its own C there)."""
        channels = x.shape[0] if channels is None else channels
        assert out.shape[1] == channels, (out.shape, channels)
        _hip.check(_hip.lib().nunif_hip_gather_tiles(_ptr(x), _ptr(out), ctypes.byref(self.grid), channels,
                                                     tile_begin, n_tiles, _hip.current_stream_ptr(x.device)))
        return out

    @staticmethod
    def tiled_render(x, model, tile_size=None, batch_size=None, enable_amp=True,
                     config_callback=None, preprocess_callback=None, input_callback=None):
        assert not torch.is_grad_enabled()
        if config_callback is None:
            C, H, W = x.shape
            output_base_shape = x.shape
        else:
            C, H, W, D = config_callback(x)
            output_base_shape = (D, *x.shape[1:])
        scale, offset = model.i2i_scale, model.i2i_offset
        blend_size = model.i2i_blend_size or 0
        batch_size = batch_size or model.i2i_default_batch_size
        tile_size = model.find_valid_tile_size(tile_size)
        device = get_model_device(model)
        if device.type != "cuda":
            raise RuntimeError("nunif_amd tiled_render needs the model on a ROCm device (no CPU fallback)")
        in_dtype = x.dtype
        x = x.to(device=device, dtype=torch.float32).contiguous()

        if (config_callback is None and preprocess_callback is None and input_callback is None
                and hasattr(model, "render_frame")):
            # whole-frame native path: gather fused into the first conv, single-pass stitch
            return model.render_frame(x, tile_size=tile_size, batch_size=batch_size).to(in_dtype)

        sb = SeamBlending(output_base_shape, scale=scale, offset=offset, tile_size=tile_size, blend_size=blend_size)
        n_tiles = sb.h_blocks * sb.w_blocks
        custom_input = preprocess_callback is not None or input_callback is not None
        if preprocess_callback is not None:
            with _device.autocast(device, enabled=enable_amp):
                x = preprocess_callback(x, sb.pad)
        elif input_callback is not None:
            x = torch.nn.functional.pad(x.unsqueeze(0), sb.pad, mode="replicate")[0]
        minibatch = torch.empty((batch_size, C, tile_size, tile_size), dtype=torch.float32, device=device)
        for t0 in range(0, n_tiles, batch_size):
            nb = min(batch_size, n_tiles - t0)
            if custom_input:
                for k in range(nb):
                    hi, wi = divmod(t0 + k, sb.w_blocks)
                    i, j = hi * sb.input_tile_step, wi * sb.input_tile_step
                    if input_callback is not None:
                        minibatch[k] = input_callback(x, i, i + tile_size, j, j + tile_size)
                    else:
                        minibatch[k] = x[:, i:i + tile_size, j:j + tile_size]
            else:
                sb.gather(x, t0, nb, minibatch, channels=C)
            with _device.autocast(device, enabled=enable_amp):
                z = model(minibatch[:nb])
            sb._store(device)[t0:t0 + nb].copy_(z)
        return sb.get_output().to(in_dtype).contiguous()

    @staticmethod
    def create_config(x_size, scale, offset, tile_size, blend_size):
        return _hip.tile_grid(x_size[0], x_size[1], scale, offset, tile_size, blend_size).as_config()

    @staticmethod
    def create_blend_filter(scale, offset, tile_size, blend_size, out_channels):
        """F[c, y, x] = min(r[y], r[x]) with the reference's fp32 ramp (:146-153)."""
        n = tile_size * scale - offset * 2
        r = torch.ones(n, dtype=torch.float32)
        if blend_size > 0:
            ramp = torch.tensor(_hip.blend_ramp(blend_size), dtype=torch.float32)
            r[:blend_size] = ramp
            r[n - blend_size:] = ramp.flip(0)
        f = torch.minimum(r[:, None], r[None, :])
        return f.unsqueeze(0).repeat(out_channels, 1, 1)
