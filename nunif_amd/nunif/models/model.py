"""Model base classes.  Mirrors ``nunif/models/model.py`` (Model :7-40, tile-size validators :43-62,
I2IBaseModel :65-86): the attribute contract ``tiled_render`` relies on.
"""
import copy
from functools import lru_cache

import torch.nn as nn


class Model(nn.Module):
    name = "nunif.Model"

    def __init__(self, kwargs):
        super().__init__()
        self.kwargs = {}
        self.updated_at = None
        self.register_kwargs(kwargs)

    def get_device(self):
        return next(self.parameters()).device

    def register_kwargs(self, kwargs):
        self.kwargs.update({k: v for k, v in kwargs.items() if k not in ("self", "__class__")})

    def get_kwargs(self):
        return self.kwargs

    def to_inference_model(self):
        return copy.deepcopy(self).eval()

    def __deepcopy__(self, memo):
        """Engine-backed models hold a ctypes handle to device state in ``_engine`` (not copyable, and a shallow copy would
        free it twice): the copy gets its own weights and NO engine — it is rebuilt lazily on the first forward."""
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_engine" else copy.deepcopy(v, memo)
        return new


_validators = {}


@lru_cache(maxsize=None)
def _largest_valid_tile(name, requested):
    check = _validators.get(name)
    if check is None:
        return int(requested)
    for size in range(int(requested), 0, -1):
        if check(size):
            return size
    raise ValueError(f"Could not find valid tile size: tile_size={requested}")


class I2IBaseModel(Model):
    """Image-to-image model: ``[B,C,T,T] -> [B,C,T*scale-2*offset, ...]`` in [0,1]."""
    name = "nunif.i2i_base_model"

    def __init__(self, kwargs, scale, offset, in_channels=None, in_size=None, blend_size=None,
                 default_tile_size=256, default_batch_size=4):
        super().__init__(kwargs)
        self.i2i_scale = scale
        self.i2i_offset = offset
        self.i2i_in_channels = in_channels
        self.i2i_in_size = in_size
        self.i2i_blend_size = blend_size
        self.i2i_default_tile_size = default_tile_size
        self.i2i_default_batch_size = default_batch_size

    def register_tile_size_validator(self, validator):
        _validators[self.name] = validator

    def find_valid_tile_size(self, base_tile_size):
        if base_tile_size is None:
            base_tile_size = self.i2i_default_tile_size
        return _largest_valid_tile(self.name, base_tile_size)
