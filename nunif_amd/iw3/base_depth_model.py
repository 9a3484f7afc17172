"""Depth model contract on the HIP engine.

Mirrors ``iw3/base_depth_model.py`` ``BaseDepthModel`` :30-238 — ``load``, abstract ``infer``, the EMA min-max
plumbing (``enable_ema`` / ``disable_ema`` / ``reset`` / ``minmax_normalize`` / ``flush_minmax_normalize``), 16-bit
depth PNG I/O — and provides ``CallableDepthModel``: the ``DepthAnythingModel.infer`` pipeline
(``iw3/depth_anything_model.py:241-253`` → ``batch_infer`` :123-182) around an arbitrary backbone callable.  The
reference's backbones live in external ``torch.hub`` repositories (:200-230) that cannot be fetched here; the in-tree HIP
ViT / DPT (``nunif_amd/iw3/depth_anything_v2.py``, SURVEY.md §8f row f2, pinned against HuggingFace since round 3) is the
backbone ``DepthAnythingModel`` puts behind this class.

Multi-GPU: the reference swaps in ``DeviceSwitchInference`` replicas for a thread pool (:129-133).  Here a model is
bound to ONE device; frames are sharded across ranks by ``nunif_amd.parallel`` instead.
"""
from abc import ABCMeta, abstractmethod

import torch

from ..nunif.device import create_device
from .depth_anything_model import batch_infer
from .depth_scaler import EMAMinMaxScaler


class BaseDepthModel(metaclass=ABCMeta):
    def __init__(self, model_type):
        self.device = None
        self.model = None
        self.model_type = model_type
        self.scaler = self.create_depth_scaler()
        self.limit_resolution = False

    def create_depth_scaler(self):
        return EMAMinMaxScaler(decay=0, buffer_size=1)

    @classmethod
    def get_name(cls):
        return cls.__name__

    def loaded(self):
        return self.model is not None

    @classmethod
    def multi_gpu_supported(cls, name):
        return False

    @abstractmethod
    def load_model(self, model_type, resolution, device):
        pass

    @abstractmethod
    def is_metric(self):
        pass

    def load(self, gpu=0, resolution=None, limit_resolution=False, **kwargs):
        self.device = create_device(gpu)
        self.limit_resolution = limit_resolution
        self.model = self.load_model(self.model_type, resolution=resolution, device=self.device, **kwargs)
        if hasattr(self.model, "to"):
            self.model = self.model.to(self.device)
        if hasattr(self.model, "eval"):
            self.model = self.model.eval()
        if isinstance(gpu, (list, tuple)) and len(gpu) > 1:
            # iw3/base_depth_model.py:129-133: one replica per listed device, the call goes to the replica of x.device
            if self.multi_gpu_supported(self.model_type):
                from ..nunif.models.data_parallel import DeviceSwitchInference
                self.model = DeviceSwitchInference(self.model, device_ids=list(gpu))
            else:
                raise ValueError(f"{self.model_type} does not support Multi-GPU")
        return self

    def get_model(self):
        return self.model

    # -- lifecycle parity (iw3/base_depth_model.py:43-49,56-96,140-149).  The engine's kernels are already native: compiling
    #    is a no-op, so the context manager the CLI wraps around video processing costs nothing.
    def compile_context(self, enabled=True):
        import contextlib
        return contextlib.nullcontext()

    def compile(self):
        pass

    def clear_compiled_model(self):
        pass

    @classmethod
    def supported(cls, model_type):
        return False

    @classmethod
    def has_checkpoint_file(cls, model_type):
        return False

    @classmethod
    def get_model_path(cls, model_type):
        return None

    @classmethod
    def force_update(cls):
        pass

    def is_image_supported(self):
        return True

    def is_video_supported(self):
        return True

    @abstractmethod
    def infer(self, x, *kwargs):          # (sic: ``*kwargs`` is the reference's own abstract signature, base_depth_model.py:151)
        pass

    # -- normalisation plumbing ---------------------------------------------------------------------------------------
    def enable_ema(self, decay, buffer_size=None):
        self.scaler.reset(decay=decay, buffer_size=buffer_size)

    def get_ema_state(self):
        return self.scaler.decay, self.scaler.buffer_size

    def disable_ema(self):
        self.scaler.reset(decay=0, buffer_size=1)

    def reset_ema(self, decay=None, buffer_size=None):
        self.scaler.reset(decay=decay, buffer_size=buffer_size)

    def reset_state(self):
        pass

    def reset(self):
        self.reset_ema()
        self.reset_state()

    def get_ema_buffer_size(self):
        return self.scaler.buffer_size

    def minmax_normalize_chw(self, depth, return_minmax=False):
        return self.scaler(depth, return_minmax=return_minmax)

    def flush_minmax_normalize(self, return_minmax=False):
        return self.scaler.flush(return_minmax=return_minmax)

    def minmax_normalize(self, depth, reset_ema=None):
        assert depth.ndim == 4
        reset_ema = [False] * depth.shape[0] if reset_ema is None else reset_ema
        assert len(reset_ema) == depth.shape[0]
        out = []
        for i in range(depth.shape[0]):
            d = self.minmax_normalize_chw(depth[i])
            if d is not None:
                out.append(d)
            if reset_ema[i]:
                out += self.flush_minmax_normalize()
                self.reset_ema()
        return out

    # -- 16-bit depth PNG ---------------------------------------------------------------------------------------------
    @staticmethod
    def save_normalized_depth(depth, file_path, png_info={}, min_depth_value=None, max_depth_value=None):
        from PIL import Image
        from PIL.PngImagePlugin import PngInfo
        info = dict(png_info)
        if min_depth_value is not None:
            info.update(iw3_min_depth_value=float(min_depth_value))
        if max_depth_value is not None:
            info.update(iw3_max_depth_value=float(max_depth_value))
        arr = (0xffff * torch.clamp(depth, 0, 1)).to(torch.int32).squeeze(0).cpu().numpy().astype("uint16")
        meta = PngInfo()
        for k, v in info.items():
            meta.add_text(k, str(v))
        Image.fromarray(arr).save(file_path, pnginfo=meta)

    @staticmethod
    def load_depth(file_path):
        import numpy as np
        from PIL import Image
        with Image.open(file_path) as im:
            text = dict(getattr(im, "text", {}))
            try:
                lo, hi = float(text["iw3_min_depth_value"]), float(text["iw3_max_depth_value"])
            except (KeyError, ValueError, TypeError):
                lo = hi = None
            arr = np.asarray(im)
            depth = torch.from_numpy(arr.astype("float32"))
            depth = depth.unsqueeze(0) if depth.dim() == 2 else depth.permute(2, 0, 1)
            if arr.dtype != np.float32:
                depth = torch.clamp(depth / 0xffff, 0, 1)
            if depth.shape[0] != 1:
                depth = depth.mean(dim=0, keepdim=True)
            if lo is not None:
                depth = depth * (hi - lo) + lo
            text["filename"] = file_path
            return depth, text


class CallableDepthModel(BaseDepthModel):
    """``DepthAnythingModel``-style wrapper around any ``backbone(x[B,3,h,w]) -> [B,h,w]`` (larger = nearer)."""

    def __init__(self, backbone, model_type="callable", metric_depth=False, lower_bound=392):
        super().__init__(model_type)
        self._backbone, self._metric, self.lower_bound = backbone, metric_depth, lower_bound

    def load_model(self, model_type, resolution=None, device=None, **kwargs):
        if resolution is not None:
            self.lower_bound = resolution
        return self._backbone

    def is_metric(self):
        return self._metric

    @torch.inference_mode()
    def infer(self, x, tta=False, low_vram=False, enable_amp=True, edge_dilation=0, depth_aa=False, **kwargs):
        if not torch.is_tensor(x):
            raise ValueError("infer expects a CHW or BCHW float tensor in [0,1]")
        aa = None
        if depth_aa:
            aa = getattr(self, "depth_aa", None)
            if aa is None:
                raise ValueError("depth_aa=True needs model.depth_aa = nunif_amd.iw3.models.DepthAA (weights loaded, on the device)")
        return batch_infer(self.model, x.to(self.device), flip_aug=tta, enable_amp=enable_amp,
                           edge_dilation=edge_dilation, lower_bound=self.lower_bound,
                           limit_resolution=self.limit_resolution, metric_depth=self._metric, depth_aa=aa)
