"""``Waifu2x`` task context on the HIP engine.

Mirrors ``waifu2x/utils.py`` (reference) :42-297 — constructor, model slots per method / noise level, ``load_model``,
``load_model_all``, ``render``, ``convert`` (alpha handling + TTA), ``to/half/float/compile/warmup`` — with the same
assertions and exceptions (AssertionError for bad arguments, FileNotFoundError for missing model files, ValueError for
an unknown method).  ``gpus`` with several ids drives one engine replica per listed device from this process (the tile minibatch is split like the
reference's ``nn.DataParallel``, ``nunif/models/data_parallel.py``); ``compile``/``half`` are no-ops because the engine is
native fp16-storage / fp32-accumulate already.
"""
from os import path

import torch
import torch.nn.functional as F

from ..nunif.device import create_device
from ..nunif.models import load_model
from ..nunif.models.register import data_parallel_model
from ..nunif.transforms.tta import tta_merge, tta_split
from ..nunif.utils.alpha import AlphaBorderPadding
from ..nunif.utils.render import tiled_render
from .models import cunet, swin_unet, swin_unet_v2, vgg_7  # noqa: F401  (registers waifu2x.swin_unet_*, swin_unet_v2_*, cunet, upcunet, vgg_7, upconv_7)

METHODS = ("scale", "scale4x", "noise_scale", "noise_scale4x", "noise")



def can_compile(model):
    """waifu2x/utils.py:25-40 decides whether ``torch.compile`` may wrap a model.  Engine models are already native."""
    return False

class Waifu2x():
    def __init__(self, model_dir, gpus):
        self.scale_model = None
        self.scale4x_model = None
        self.noise_models = [None] * 4
        self.noise_scale_models = [None] * 4
        self.noise_scale4x_models = [None] * 4
        self.device = create_device(gpus)
        self.gpus = gpus
        self.model_dir = model_dir
        self.alpha_pad = AlphaBorderPadding()
        self.is_half = False

    # -- lifecycle ------------------------------------------------------------------------------------------------
    def _models(self):
        return [m for m in (self.scale_model, self.scale4x_model, *self.noise_models, *self.noise_scale_models,
                            *self.noise_scale4x_models) if m is not None]

    def _apply(self, func):
        slot = lambda m: func(m) if m is not None else None     # noqa: E731
        self.scale_model = slot(self.scale_model)
        self.scale4x_model = slot(self.scale4x_model)
        self.noise_models = [slot(m) for m in self.noise_models]
        self.noise_scale_models = [slot(m) for m in self.noise_scale_models]
        self.noise_scale4x_models = [slot(m) for m in self.noise_scale4x_models]

    def _setup(self):
        self._apply(lambda model: model.to(self.device).eval())

    def compile(self):
        return self

    def to(self, device):
        self.device = torch.device(device)
        self._setup()
        return self

    def half(self):
        self.is_half = True
        return self

    def float(self):
        self.is_half = False
        return self

    @torch.inference_mode()
    def warmup(self, tile_size, batch_size, enable_amp):
        for model in self._models():
            t = model.find_valid_tile_size(tile_size)
            x = torch.zeros((batch_size or model.i2i_default_batch_size, 3, t, t), device=self.device)
            model(x)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    # -- loading --------------------------------------------------------------------------------------------------
    def has_model_file(self, filename):
        return path.exists(path.join(self.model_dir, filename))

    def load_model_by_name(self, filename):
        return load_model(path.join(self.model_dir, filename), map_location="cpu", device_ids=self.gpus,
                          weights_only=True)[0]

    def _dp(self, model):
        """waifu2x/utils.py:144,163,174: a model derived from the 4x net is wrapped for the listed devices like a loaded one."""
        gpus = self.gpus if isinstance(self.gpus, (list, tuple)) else [self.gpus]
        return data_parallel_model(model, device_ids=list(gpus)) if len(gpus) > 1 else model

    # Where a method's model comes from: its own file, or — when that file is absent — a model DERIVED from the 4x net of the
    # same noise level (``SwinUNet4x.to_2x()`` / ``to_1x()``: the 4x output downscaled, reference :139-175).  One table instead of
    # the reference's five hand-written branches; the slots themselves keep the reference's attribute names.
    _FILE = {"scale": "scale2x.pth", "scale4x": "scale4x.pth", "noise": "noise{n}.pth",
             "noise_scale": "noise{n}_scale2x.pth", "noise_scale4x": "noise{n}_scale4x.pth"}
    _DERIVED_FROM = {"scale": ("scale4x", "to_2x"), "noise_scale": ("noise_scale4x", "to_2x"), "noise": ("noise_scale4x", "to_1x")}
    _ATTR = {"scale": "scale_model", "scale4x": "scale4x_model", "noise": "noise_models", "noise_scale": "noise_scale_models",
             "noise_scale4x": "noise_scale4x_models"}

    def _slot(self, method, noise_level):
        held = getattr(self, self._ATTR[method])
        return held[noise_level] if isinstance(held, list) else held

    def _fill_slot(self, method, noise_level, model):
        if isinstance(getattr(self, self._ATTR[method]), list):
            getattr(self, self._ATTR[method])[noise_level] = model
        else:
            setattr(self, self._ATTR[method], model)

    def _load_model(self, method, noise_level):
        if method not in self._FILE:
            raise ValueError(method)
        if self._slot(method, noise_level) is not None:
            return
        filename = self._FILE[method].format(n=noise_level)
        if self.has_model_file(filename):
            model = self.load_model_by_name(filename)
        elif method in self._DERIVED_FROM:
            parent, derive = self._DERIVED_FROM[method]
            self._load_model(parent, noise_level)                 # raises FileNotFoundError when the 4x file is missing too
            model = self._dp(getattr(self._slot(parent, noise_level), derive)())
        else:
            raise FileNotFoundError(f"{filename} not found in {self.model_dir}")
        self._fill_slot(method, noise_level, model)

    def load_model(self, method, noise_level):
        assert method in METHODS
        assert method in {"scale", "scale4x"} or 0 <= noise_level and noise_level < 4
        self._load_model(method, noise_level)
        # the plain scale model of the same factor upscales the alpha channel (reference :188-203)
        companion = {"noise_scale4x": "scale4x", "noise_scale": "scale"}.get(method)
        if companion:
            try:
                self._load_model(companion, -1)
            except FileNotFoundError:
                pass    # alpha falls back to bilinear
        self._setup()

    def load_model_all(self, load_4x=True):
        if load_4x:
            self._load_model("scale4x", -1)
            for n in range(4):
                self._load_model("noise_scale4x", n)
        self._load_model("scale", -1)
        for n in range(4):
            self._load_model("noise_scale", n)
            self._load_model("noise", n)
        if not load_4x:
            self.scale4x_model = None
            self.noise_scale4x_models = [None] * 4
        self._setup()

    # -- inference ------------------------------------------------------------------------------------------------
    def render(self, x, method, noise_level, tile_size=None, batch_size=None, enable_amp=False):
        assert method in METHODS
        assert method in {"scale", "scale4x"} or 0 <= noise_level and noise_level < 4
        return tiled_render(x, self._slot(method, noise_level), tile_size=tile_size, batch_size=batch_size,
                            enable_amp=enable_amp)

    def _model_offset(self, method, noise_level):
        return self._slot(method, noise_level).i2i_offset

    def convert(self, x, alpha, method, noise_level, tile_size=None, batch_size=None, tta=False, enable_amp=False,
                output_device="cpu"):
        assert not torch.is_grad_enabled()
        assert x.shape[0] == 3
        assert alpha is None or alpha.shape[0] == 1 and alpha.shape[1:] == x.shape[1:]
        assert method in METHODS
        assert method in {"scale", "scale4x"} or 0 <= noise_level and noise_level < 4
        x = x.to(self.device)
        blank_alpha = True
        if alpha is not None:
            blank_alpha = bool(torch.all(alpha == 1))
            if not blank_alpha:
                alpha = alpha.to(self.device)
                x = self.alpha_pad(x, alpha, self._model_offset(method, noise_level))
        if tta:
            rgb = tta_merge([self.render(v, method, noise_level, tile_size, batch_size, enable_amp)
                             for v in tta_split(x)])
        else:
            rgb = self.render(x, method, noise_level, tile_size, batch_size, enable_amp)
        rgb = rgb.to(output_device)
        if alpha is not None and method != "noise":
            factor = 4 if method in {"scale4x", "noise_scale4x"} else 2
            if blank_alpha:
                alpha = F.interpolate(alpha.unsqueeze(0), scale_factor=factor, mode="nearest").squeeze(0)
            else:
                model = self.scale4x_model if factor == 4 else self.scale_model
                if model is not None:
                    a3 = alpha.expand(3, alpha.shape[1], alpha.shape[2])
                    alpha = tiled_render(a3, model, tile_size=tile_size, batch_size=batch_size,
                                         enable_amp=enable_amp).mean(0, keepdim=True)
                else:
                    alpha = F.interpolate(alpha.unsqueeze(0), scale_factor=factor, mode="bilinear").squeeze(0)
            alpha = alpha.to(output_device)
        return rgb, alpha
