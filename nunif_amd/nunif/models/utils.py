"""``.pth`` container I/O and small helpers.  Mirrors ``nunif/models/utils.py`` save_model :15-39,
load_model :42-74, get_model_kwargs :77-84, get_model_device :87-93.

File format (kept byte-compatible): ``torch.save({"nunif_model": 1, "name", "updated_at", "kwargs",
"train_kwargs", "state_dict"})``.
"""
from datetime import datetime, timezone

import torch

from .model import Model
from .register import create_model
from ..device import create_device


def save_model(model, model_path, updated_at=None, train_kwargs=None, **kwargs):
    assert isinstance(model, Model)
    if train_kwargs is not None and not isinstance(train_kwargs, dict):
        train_kwargs = vars(train_kwargs)
    if train_kwargs is not None:
        train_kwargs = {k: v for k, v in train_kwargs.items() if not callable(v)}
    data = {"nunif_model": 1, "name": model.name, "updated_at": str(updated_at or datetime.now(timezone.utc)),
            "kwargs": model.get_kwargs(), "train_kwargs": train_kwargs, "state_dict": model.state_dict()}
    data.update(kwargs)
    torch.save(data, model_path)


def load_model(model_path, model=None, device_ids=None, strict=True, map_location="cpu", weights_only=False):
    if model_path.startswith(("http://", "https://")):
        data = torch.hub.load_state_dict_from_url(model_path, weights_only=True, map_location=map_location)
    else:
        data = torch.load(model_path, map_location=map_location, weights_only=weights_only)
    assert "nunif_model" in data
    predefined = model is not None
    if not predefined:
        model = create_model(data["name"], **data["kwargs"])
    model.load_state_dict(data["state_dict"], strict=strict)
    if "updated_at" in data:
        model.updated_at = data["updated_at"]
    data.pop("state_dict")
    if not predefined and device_ids is not None:
        if len(device_ids) > 1:
            # the reference creates the DataParallel wrapper first and loads the weights into wrapper.module (:57-66); an
            # engine replica is a copy of the LOADED weights, so the order is: load, then replicate over the devices
            from .register import data_parallel_model
            model = data_parallel_model(model, device_ids)
        else:
            model = model.to(create_device(device_ids))
    return model, data


def get_model_kwargs(model, key=None):
    kwargs = model.get_kwargs()
    return kwargs if key is None else kwargs[key]


def get_model_device(model):
    from .data_parallel import DataParallelInference
    if isinstance(model, DataParallelInference):
        return model.output_device
    if hasattr(model, "get_device"):
        return model.get_device()
    return next(model.parameters()).device


def compile_model(model, **kwargs):
    """The reference gates ``torch.compile`` here (:123-132).  HIP-engine models are already native."""
    return model


def is_compiled_model(model):
    return True
