"""Public image API on the HIP engine.  Mirrors ``waifu2x/hub.py`` ``Waifu2xImageModel`` :31-163 and the ``waifu2x``
factory :166-175 (same argument names, method normalisation and ValueErrors).  Model files are looked up under
``model_dir`` (the reference downloads them into ``waifu2x/pretrained_models``; there is no network here, so the
directory must be passed or ``NUNIF_WAIFU2X_MODEL_DIR`` set)."""
import os
from os import path

import torch

from .utils import Waifu2x

MODEL_ARCH_DIRS = {
    "art": ("swin_unet", "art"), "art_scan": ("swin_unet", "art_scan"), "photo": ("swin_unet", "photo"),
    "swin_unet/art": ("swin_unet", "art"), "swin_unet/art_scan": ("swin_unet", "art_scan"),
    "swin_unet/photo": ("swin_unet", "photo"),
}
METHODS = ["noise", "scale", "noise_scale", "scale2x", "noise_scale2x", "scale4x", "noise_scale4x"]


def default_model_root():
    return os.environ.get("NUNIF_WAIFU2X_MODEL_DIR", path.join(path.dirname(__file__), "pretrained_models"))


class Waifu2xImageModel():
    def __init__(self, model_type, method=None, noise_level=-1, device_ids=[-1], tile_size=None, batch_size=None,
                 keep_alpha=True, amp=True, model_root=None):
        if model_type not in MODEL_ARCH_DIRS:
            raise ValueError(f"model_type: choose from {list(MODEL_ARCH_DIRS.keys())}")
        if method is not None and method not in METHODS:
            raise ValueError(f"method: choose from {METHODS}")
        if method is not None and method.startswith("noise") and noise_level not in {0, 1, 2, 3}:
            raise ValueError("noise_level: choose from [0, 1, 2, 3]")
        self.model_type, self.tile_size, self.batch_size = model_type, tile_size, batch_size
        self.keep_alpha, self.amp = keep_alpha, amp
        self.ctx = Waifu2x(path.join(model_root or default_model_root(), *MODEL_ARCH_DIRS[model_type]), device_ids)
        if method is not None:
            method = self.normalize_method(method, noise_level)
            self.ctx.load_model(method, noise_level)
            self.set_mode(method, noise_level)
        else:
            self.method = self.noise_level = None
            self.ctx.load_model_all(load_4x=True)

    def set_mode(self, method, noise_level=-1):
        method = self.normalize_method(method, noise_level)
        if method in {"noise", "noise_scale4x", "noise_scale"} and noise_level not in {0, 1, 2, 3}:
            raise ValueError("noise_level: choose from (0, 1, 2, 3)")
        self.method, self.noise_level = method, noise_level

    def compile(self):
        return self

    def to(self, device):
        self.ctx = self.ctx.to(device)
        return self

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        """hub.py:80-81.  The model can be parked on the CPU (weights only); ``infer`` raises there: no CPU fallback."""
        return self.to("cpu")

    def __call__(self, x, tta=False, output_type="pil", **kwargs):
        """hub.py:163: ``model(x)`` == ``model.infer(x)``."""
        return self.infer(x, tta=tta, output_type=output_type, **kwargs)

    def half(self):
        self.ctx.half()
        return self

    def float(self):
        self.ctx.float()
        return self

    @property
    def is_half(self):
        return self.ctx.is_half

    @property
    def device(self):
        return self.ctx.device

    def infer_tensor(self, rgb, alpha=None, tta=False, output_type="pil", **kwargs):
        method = kwargs.get("method", self.method)
        noise_level = kwargs.get("noise_level", self.noise_level)
        if method is None:
            raise ValueError("method is None. Call `model.set_mode(method, noise_level)` or use method and "
                             "noise_level kwargs")
        with torch.inference_mode():
            rgb, alpha = self.ctx.convert(rgb, alpha, method, noise_level, tile_size=self.tile_size,
                                          batch_size=self.batch_size, tta=tta, enable_amp=self.amp)
        if output_type == "tensor":
            return rgb, alpha
        return _to_pil(rgb, alpha)

    def infer_pil(self, pil_image, tta=False, output_type="pil", **kwargs):
        rgb, alpha = _from_pil(pil_image, self.keep_alpha)
        return self.infer_tensor(rgb.to(self.device), None if alpha is None else alpha.to(self.device), tta=tta,
                                 output_type=output_type, **kwargs)

    def infer_file(self, filepath, tta=False, output_type="pil", **kwargs):
        from PIL import Image
        return self.infer_pil(Image.open(filepath), tta=tta, output_type=output_type, **kwargs)

    def convert(self, input_filepath, output_filepath, tta=False, format="png", **kwargs):
        self.infer_file(input_filepath, tta=tta, **kwargs).save(output_filepath, format=format)

    def infer(self, x, tta=False, output_type="pil", **kwargs):
        if isinstance(x, str):
            return self.infer_file(x, tta=tta, output_type=output_type, **kwargs)
        if torch.is_tensor(x):
            return self.infer_tensor(x, tta=tta, output_type=output_type, **kwargs)
        if hasattr(x, "convert") and hasattr(x, "size"):      # PIL.Image.Image
            return self.infer_pil(x, tta=tta, output_type=output_type, **kwargs)
        raise ValueError("Unsupported input format")

    @staticmethod
    def normalize_method(method, noise_level):
        if method is None:
            return None
        method = {"scale2x": "scale", "noise_scale2x": "noise_scale"}.get(method, method)
        if method == "scale" and noise_level >= 0:
            method = "noise_scale"
        if method == "scale4x" and noise_level >= 0:
            method = "noise_scale4x"
        return method


def _from_pil(im, keep_alpha):
    import numpy as np
    has_alpha = im.mode in ("RGBA", "LA") or "transparency" in im.info
    arr = torch.from_numpy(np.asarray(im.convert("RGBA" if has_alpha else "RGB")).copy()).permute(2, 0, 1)
    x = arr.float() / 255.0
    if has_alpha and keep_alpha:
        return x[:3].contiguous(), x[3:4].contiguous()
    if has_alpha:
        # nunif/utils/pil_io.py remove_alpha (:26-30) via _load_image(keep_alpha=False): transparent pixels are composited
        # onto a WHITE background (bg_color = 255) with PIL's own integer blend — they do not just lose their alpha channel
        from PIL import Image
        rgba = im.convert("RGBA")
        nobg = Image.new("RGB", rgba.size, (255, 255, 255))
        nobg.paste(rgba, rgba.getchannel("A"))
        x = torch.from_numpy(np.asarray(nobg).copy()).permute(2, 0, 1).float() / 255.0
    return x[:3].contiguous(), None


def _to_pil(rgb, alpha):
    from PIL import Image
    x = rgb if alpha is None else torch.cat([rgb, alpha.to(rgb.device)], 0)
    # quantize256 (nunif/transforms/functional.py:9-12): x*255, round, clamp, uint8
    q = torch.clamp(torch.round(x.float().cpu() * 255.0), 0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    return Image.fromarray(q, "RGB" if alpha is None else "RGBA")


def waifu2x(model_type="art", method=None, noise_level=-1, device_ids=[-1], tile_size=None, batch_size=None,
            keep_alpha=True, amp=True, **kwargs):
    return Waifu2xImageModel(model_type=model_type, method=method, noise_level=noise_level, device_ids=device_ids,
                             tile_size=tile_size, batch_size=batch_size, keep_alpha=keep_alpha, amp=amp, **kwargs)
