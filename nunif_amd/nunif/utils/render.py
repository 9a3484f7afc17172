"""``tiled_render`` / ``simple_render``.  Mirrors ``nunif/utils/render.py`` :8-39."""
import math

import torch.nn.functional as F

from .seam_blending import SeamBlending
from ..device import autocast
from ..models.utils import get_model_device


def tiled_render(x, model, tile_size=None, batch_size=None, enable_amp=False):
    callbacks = {}
    if hasattr(model, "has_callback") and model.has_callback():
        callbacks = {k: getattr(model, k, None) for k in ("config_callback", "preprocess_callback", "input_callback")}
    return SeamBlending.tiled_render(x, model, tile_size=tile_size, batch_size=batch_size,
                                     enable_amp=enable_amp, **callbacks)


def simple_render(x, model, enable_amp=False, offset=None):
    scale = model.i2i_scale
    offset = model.i2i_offset if offset is None else offset
    device = get_model_device(model)
    single = x.dim() == 3
    x = (x.unsqueeze(0) if single else x).to(device)
    if offset > 0:
        x = F.pad(x, (math.ceil(offset / scale),) * 4, mode="replicate")
    with autocast(device, enabled=enable_amp):
        z = model(x)
    return z.squeeze(0) if single else z
