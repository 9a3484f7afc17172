"""``--method`` -> side model.  Mirrors ``iw3/stereo_model_factory.py`` (``get_mlbw_divergence_level`` :36-42,
``load_mlbw_model`` :45-94, ``load_row_flow_model`` :97-112, ``create_stereo_model`` :115-138) and the inpaint model table of
``iw3/inpaint_utils.py`` :34-96.

The reference fetches the ``.pth`` containers by URL into ``$NUNIF_HOME/iw3/pretrained_models/hub/checkpoints``; there is no
network here, so the same FILE NAMES are looked up in ``model_dir`` (default: that directory, ``NUNIF_HOME`` from the
environment, else ``~/.nunif``; ``<model_dir>/checkpoints/<file>`` or ``<model_dir>/<file>``) and a missing file raises
``FileNotFoundError`` naming it.  The containers are the reference's own (``nunif_amd.nunif.models.load_model`` reads them);
the models they name are the HIP-engine classes registered under the same names."""
import os

from ..nunif.models import load_model
from .forward_inpaint import ForwardInpaint
from .mlbw_inpaint import MLBWInpaint

ROW_FLOW_V3 = "iw3_row_flow_v3_20250627.pth"
ROW_FLOW_V3_SYM = "iw3_row_flow_v3_sym_20250628.pth"
MLBW = {("l2", 1): "iw3_mlbw_l2_d1_20250627.pth", ("l2", 2): "iw3_mlbw_l2_d2_20250627.pth",
        ("l2", 3): "iw3_mlbw_l2_d3_20250627.pth", ("l4", 1): "iw3_mlbw_l4_d1_20250627.pth",
        ("l4", 2): "iw3_mlbw_l4_d2_20250627.pth", ("l4", 3): "iw3_mlbw_l4_d3_20250627.pth",
        ("l2s", 1): "iw3_mlbw_l2s_d1_20250627.pth", ("l4s", 1): "iw3_mlbw_l4s_d1_20250627.pth"}
MLBW_WEAK = {("l2", 2): "iw3_mlbw_l2_d2_weak_20250627.pth", ("l2", 3): "iw3_mlbw_l2_d3_weak_20250627.pth",
             ("l4", 2): "iw3_mlbw_l4_d2_weak_20250627.pth", ("l4", 3): "iw3_mlbw_l4_d3_weak_20250627.pth"}
MASK_MLBW_L2_D1 = "iw3_mask_mlbw_l2_d1_20250903.pth"
INPAINT_MODEL_DEFAULT = "light_inpaint_v1"
INPAINT_MODELS = {INPAINT_MODEL_DEFAULT: {"video": "iw3_light_video_inpaint_v1_20250919.pth",
                                          "image": "iw3_light_inpaint_v1_20250919.pth"}}


def default_model_dir():
    home = os.environ.get("NUNIF_HOME") or os.path.join(os.path.expanduser("~"), ".nunif")
    return os.path.join(home, "iw3", "pretrained_models", "hub")


def resolve(filename, model_dir=None):
    if os.path.isabs(filename) and os.path.exists(filename):
        return filename
    model_dir = model_dir or default_model_dir()
    for p in (os.path.join(model_dir, "checkpoints", filename), os.path.join(model_dir, filename)):
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"{filename} not found under {model_dir} (no downloads here: copy the reference's checkpoint "
                            f"there; it is https://github.com/nagadomi/nunif/releases/download/0.0.0/{filename})")


def _load(filename, device_id, model_dir):
    return load_model(resolve(filename, model_dir), weights_only=True, device_ids=[device_id])[0].eval()


def get_mlbw_divergence_level(d):
    return 1 if d <= 4 else (2 if d <= 7 else 3)


def load_mlbw_model(method, divergence, device_id, use_weak_convergence_model=False, model_dir=None):
    level = get_mlbw_divergence_level(divergence)
    if method in {"mlbw_l2", "mlbw_l2s", "mlbw_l4", "mlbw_l4s"}:
        kind = method[len("mlbw_"):]
        if level == 1:
            filename = MLBW[(kind, 1)]
        else:                                       # the small nets only exist for level 1: l2s -> l2, l4s -> l4 above it
            table = MLBW_WEAK if use_weak_convergence_model else MLBW
            filename = table[(kind.rstrip("s"), level)]
    elif method == "mask_mlbw_l2":
        filename = MASK_MLBW_L2_D1
    else:
        raise ValueError(method)
    model = _load(filename, device_id, model_dir)
    model.delta_output = True
    return model


def load_row_flow_model(method, device_id, model_dir=None):
    if method in {"row_flow_v3", "row_flow"}:
        model = _load(ROW_FLOW_V3, device_id, model_dir)
        model.symmetric = False
    elif method in {"row_flow_v3_sym", "row_flow_sym"}:
        model = _load(ROW_FLOW_V3_SYM, device_id, model_dir)
        model.symmetric = True
    elif method == "row_flow_v2":
        raise NotImplementedError("row_flow_v2 (the 2024 legacy net) is not on the HIP engine; use row_flow_v3")
    else:
        raise ValueError(method)
    model.delta_output = True
    return model


def _inpaint_files(name):
    name = name or INPAINT_MODEL_DEFAULT
    if name not in INPAINT_MODELS:
        raise ValueError(f"inpaint model `{name}` is not defined")
    return INPAINT_MODELS[name]


def load_image_inpaint_model(name, device_id, model_dir=None):
    return _load(_inpaint_files(name)["image"], device_id, model_dir)


def load_video_inpaint_model(name, device_id, model_dir=None):
    return _load(_inpaint_files(name)["video"], device_id, model_dir)


def create_stereo_model(method, divergence, device_id, use_weak_convergence_model=False, inpaint_model=None, model_dir=None):
    if method.startswith("row_flow"):
        return load_row_flow_model(method, device_id=device_id, model_dir=model_dir)
    if method == "mlbw_l2_inpaint":
        return MLBWInpaint(load_image_inpaint_model(inpaint_model, device_id, model_dir),
                           load_mlbw_model("mask_mlbw_l2", divergence, device_id, model_dir=model_dir),
                           video_model=load_video_inpaint_model(inpaint_model, device_id, model_dir))
    if method.startswith("mlbw_") or method.startswith("mask_mlbw_"):
        return load_mlbw_model(method, divergence=divergence, device_id=device_id,
                               use_weak_convergence_model=use_weak_convergence_model, model_dir=model_dir)
    if method in {"forward", "forward_fill", "backward", "grid_sample", "NULL"}:
        return None
    if method == "forward_inpaint":
        return ForwardInpaint(load_image_inpaint_model(inpaint_model, device_id, model_dir),
                              video_model=load_video_inpaint_model(inpaint_model, device_id, model_dir))
    raise ValueError(method)
