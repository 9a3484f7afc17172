"""Name -> model factory registry.  Mirrors ``nunif/models/register.py`` :9-68.

Multi-GPU: the reference wraps the model in ``nn.DataParallel`` over the tile minibatch (:44-49).  This engine
shards whole frames across one-process-per-GPU ranks instead (``nunif_amd.parallel``), so ``device_ids`` with
more than one entry binds the model to the first id.
"""
import inspect

from .model import Model
from ..device import create_device

_models = {}


def register_model(cls):
    assert issubclass(cls, Model)
    _models[cls.name] = cls
    for alias in getattr(cls, "name_alias", ()):
        _models[alias] = cls
    return cls


def register_model_factory(name, func):
    _models[name] = func


def data_parallel_model(model, device_ids):
    return model


def create_model(name, device_ids=None, **kwargs):
    if name not in _models:
        raise ValueError(f"Unknown model name: {name}")
    model = _models[name](**kwargs)
    if device_ids is not None:
        model = model.to(create_device(device_ids))
    return model


def get_model_names():
    return list(_models.keys())


def register_models(module):
    for _, obj in inspect.getmembers(module, inspect.isclass):
        if issubclass(obj, Model) and obj is not Model:
            register_model(obj)
