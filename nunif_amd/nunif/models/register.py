"""Name -> model factory registry.  Mirrors ``nunif/models/register.py`` :9-68.

Multi-GPU: the reference wraps the model in ``nn.DataParallel`` over the tile minibatch (:44-49); ``device_ids`` with more
than one entry does the same here through ``data_parallel.DataParallelWrapper`` (one engine replica + stream per listed
device, driven from one process).  One process per GPU with whole-frame sharding (``nunif_amd.parallel``) is the faster
layout and what bench.py measures.
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
    """register.py:44-49: wrap once when more than one device is listed."""
    from .data_parallel import DataParallelWrapper
    if len(device_ids) > 1 and not isinstance(model, DataParallelWrapper):
        model = DataParallelWrapper(model, device_ids=device_ids)
    return model


def create_model(name, device_ids=None, **kwargs):
    if name not in _models:
        raise ValueError(f"Unknown model name: {name}")
    model = _models[name](**kwargs)
    if device_ids is not None:
        if len(device_ids) > 1:                      # register.py:56-61
            model = data_parallel_model(model, device_ids)
        else:
            model = model.to(create_device(device_ids))
    return model


def get_model_names():
    return list(_models.keys())


def register_models(module):
    for _, obj in inspect.getmembers(module, inspect.isclass):
        if issubclass(obj, Model) and obj is not Model:
            register_model(obj)
