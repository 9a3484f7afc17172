"""Depth models by name.  Mirrors ``iw3/depth_anything_model.py`` ``DepthAnythingModel`` :185-290 (name tables :12-66) and
``iw3/null_depth_model.py`` :1-79.

``DepthAnythingModel("Any_V2_S")`` reads the PUBLISHED checkpoint (``depth_anything_v2_vits.pth``, key layout ``pretrained.*`` /
``depth_head.*``) from ``<model_dir>/checkpoints`` — the reference lets ``torch.hub`` fetch repository and weights — and runs it
on the engine's ViT + DPT kernels (``HipDepthAnythingV2``; parity unpinned).  Every name of the reference's table
maps onto that engine: the geometry (ViT-S / B / L) comes from the checkpoint itself; what the hub entry points decide from the
name is set here — Depth-Anything V1 (``Any_S/B/L``) feeds the DPT head from the last four encoder blocks, the V2 metric
models (``Any_V2_N*`` hypersim, ``Any_V2_K*`` vkitti) end in Sigmoid x max_depth (20 / 80), ``Distill_Any_*`` are V2
geometries with their own weights."""
import os

import torch
import torch.nn.functional as F

from .base_depth_model import BaseDepthModel
from .depth_anything_model import batch_infer
from .dilation import dilate_edge, edge_dilation_is_enabled
from .stereo_model_factory import default_model_dir

NAME_MAP = {
    "Any_S": "vits", "Any_B": "vitb", "Any_L": "vitl", "Any_V2_S": "v2_vits", "Any_V2_B": "v2_vitb", "Any_V2_L": "v2_vitl",
    "Any_V2_N_S": "hypersim_s", "Any_V2_N_B": "hypersim_b", "Any_V2_N_L": "hypersim_l",
    "Any_V2_K_S": "vkitti_s", "Any_V2_K_B": "vkitti_b", "Any_V2_K_L": "vkitti_l",
    "Any_V2_N": "hypersim_l", "Any_V2_K": "vkitti_l",
    "Distill_Any_S": "distill_any_depth_s", "Distill_Any_B": "distill_any_depth_b", "Distill_Any_L": "distill_any_depth_l",
}
MODEL_FILE_NAMES = {
    "Any_S": "depth_anything_vits14.pth", "Any_B": "depth_anything_vitb14.pth", "Any_L": "depth_anything_vitl14.pth",
    "Any_V2_S": "depth_anything_v2_vits.pth", "Any_V2_B": "depth_anything_v2_vitb.pth", "Any_V2_L": "depth_anything_v2_vitl.pth",
    "Any_V2_N_S": "depth_anything_v2_metric_hypersim_vits.pth", "Any_V2_N_B": "depth_anything_v2_metric_hypersim_vitb.pth",
    "Any_V2_N_L": "depth_anything_v2_metric_hypersim_vitl.pth", "Any_V2_K_S": "depth_anything_v2_metric_vkitti_vits.pth",
    "Any_V2_K_B": "depth_anything_v2_metric_vkitti_vitb.pth", "Any_V2_K_L": "depth_anything_v2_metric_vkitti_vitl.pth",
    "Any_V2_N": "depth_anything_v2_metric_hypersim_vitl.pth", "Any_V2_K": "depth_anything_v2_metric_vkitti_vitl.pth",
    "Distill_Any_S": "distill_any_depth_vits.safetensors", "Distill_Any_B": "distill_any_depth_vitb.safetensors",
    "Distill_Any_L": "distill_any_depth_vitl.safetensors",
}
AA_SUPPORTED_MODELS = {"Any_V2_S", "Any_V2_B", "Any_V2_L"}
ENGINE_MODELS = set(MODEL_FILE_NAMES)
V1_MODELS = {"Any_S", "Any_B", "Any_L"}                  # DPT_DINOv2: get_intermediate_layers(x, 4) = the last four blocks


def head_options(model_type, n_blocks):
    """-> (taps, max_depth) of a Depth-Anything name: what ``hubconf.DepthAnything`` / ``DepthAnythingMetricDepthV2`` configure."""
    taps = tuple(range(n_blocks - 4, n_blocks)) if model_type in V1_MODELS else None
    enc = NAME_MAP[model_type]
    max_depth = 20.0 if enc.startswith("hypersim") else (80.0 if enc.startswith("vkitti") else 0.0)
    return taps, max_depth

DEPTH_AA_FILE = "iw3_depth_aa_20250530.pth"


class DepthAnythingModel(BaseDepthModel):
    def __init__(self, model_type, model_dir=None):
        super().__init__(model_type)
        self.model_dir = model_dir
        self.depth_aa = None
        self.lower_bound = 392

    @classmethod
    def _path(cls, model_type, model_dir=None):
        return os.path.join(model_dir or default_model_dir(), "checkpoints", MODEL_FILE_NAMES[model_type])

    def load_model(self, model_type, resolution=None, device=None, state_dict=None, depth_aa=None, **kwargs):
        """``state_dict`` / ``depth_aa`` let a caller hand over weights it already holds (tests, the benches)."""
        from .depth_anything_v2 import HipDepthAnythingV2
        if model_type not in ENGINE_MODELS:
            raise NotImplementedError(f"{model_type}: not a Depth-Anything name ({sorted(ENGINE_MODELS)})")
        if state_dict is None:
            p = self._path(model_type, self.model_dir)
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p} not found (no downloads here: copy the published checkpoint there)")
            if p.endswith(".safetensors"):
                from safetensors.torch import load_file
                state_dict = load_file(p)
            else:
                state_dict = torch.load(p, map_location="cpu", weights_only=True)
        self.depth_aa = depth_aa
        if depth_aa is None and model_type in AA_SUPPORTED_MODELS:
            p = os.path.join(self.model_dir or default_model_dir(), "checkpoints", DEPTH_AA_FILE)
            if os.path.exists(p):                        # optional: only needed for infer(depth_aa=True)
                from ..nunif.models import load_model
                self.depth_aa = load_model(p, weights_only=True)[0].eval().to(device)
        self.lower_bound = resolution or 392
        if self.lower_bound % 14 != 0:
            self.lower_bound += 14 - self.lower_bound % 14          # from the GUI: 512 -> 518 (:228-230)
        n_blocks = sum(1 for k in state_dict if k.startswith("pretrained.blocks.") and k.endswith(".attn.qkv.weight"))
        taps, max_depth = head_options(model_type, n_blocks)
        model = HipDepthAnythingV2(state_dict, device, taps=taps, max_depth=max_depth)
        model.prep_lower_bound = self.lower_bound
        return model

    @torch.inference_mode()
    def infer(self, x, tta=False, low_vram=False, enable_amp=True, edge_dilation=0, depth_aa=False, **kwargs):
        if not torch.is_tensor(x):
            raise ValueError("infer expects a CHW or BCHW float tensor in [0,1]")
        if depth_aa and self.depth_aa is None:
            raise ValueError(f"depth_aa=True needs {DEPTH_AA_FILE} next to the depth checkpoint (or load(depth_aa=model))")
        if x.device.type != "cuda":
            x = x.to(self.device)          # (reference :241-253: a tensor stays on ITS device — that is what picks the replica)
        return batch_infer(self.model, x, flip_aug=tta, enable_amp=enable_amp, edge_dilation=edge_dilation,
                           lower_bound=self.lower_bound, limit_resolution=self.limit_resolution,
                           metric_depth=self.is_metric(), depth_aa=self.depth_aa if depth_aa else None)

    @classmethod
    def get_name(cls):
        return "DepthAnything"

    @classmethod
    def supported(cls, model_type):
        return model_type in MODEL_FILE_NAMES

    @classmethod
    def has_checkpoint_file(cls, model_type):
        return cls.supported(model_type) and os.path.exists(cls._path(model_type))

    @classmethod
    def get_model_path(cls, model_type):
        return cls._path(model_type)

    def is_metric(self):
        return self.model_type.startswith("Any_V2_N") or self.model_type.startswith("Any_V2_K")

    @classmethod
    def multi_gpu_supported(cls, model_type):
        return True        # load(gpu=[...]): one engine per listed device (DeviceSwitchInference); or one process per GPU

    @classmethod
    def force_update(cls):
        pass


class NullDepthModel(BaseDepthModel):
    """Dummy depth model of the reference's throughput benchmarks (``--depth-model NULL``): square bilinear resize + channel
    mean.  Those two ops are ATen-on-ROCm exactly as in the reference (it is a stand-in that measures everything BUT the depth
    net); edge dilation goes through the engine."""

    def __init__(self, model_type="NULL"):
        super().__init__(model_type)
        self.resolution = 392

    def load_model(self, model_type, resolution=None, device=None, **kwargs):
        self.resolution = resolution or 392
        return lambda x: F.interpolate(x, size=(self.resolution, self.resolution), mode="bilinear").mean(dim=1, keepdim=True)

    @torch.inference_mode()
    def infer(self, x, tta=False, low_vram=False, enable_amp=True, edge_dilation=0, **kwargs):
        single = x.ndim == 3
        x = x.unsqueeze(0) if single else x
        x = (self.model(x) + self.model(x)) * 0.5 if tta else self.model(x)
        if edge_dilation_is_enabled(edge_dilation):
            x = dilate_edge(x, edge_dilation)
        return x.squeeze(0) if single else x

    @classmethod
    def get_name(cls):
        return "NullDepth"

    @classmethod
    def supported(cls, model_type):
        return model_type == "NULL"

    @classmethod
    def has_checkpoint_file(cls, model_type):
        return cls.supported(model_type)

    @classmethod
    def get_model_path(cls, model_type):
        return None

    def is_metric(self):
        return False

    def is_video_supported(self):
        return True

    @classmethod
    def multi_gpu_supported(cls, model_type):
        return True

    @classmethod
    def force_update(cls):
        pass
