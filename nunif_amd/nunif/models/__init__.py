from .model import Model, I2IBaseModel
from .register import (register_model, register_model_factory, register_models, create_model,
                       get_model_names, data_parallel_model)
from .utils import (load_model, save_model, get_model_kwargs, get_model_device, compile_model,
                    is_compiled_model)

__all__ = ["Model", "I2IBaseModel", "register_model", "register_model_factory", "register_models",
           "create_model", "get_model_names", "data_parallel_model", "load_model", "save_model",
           "get_model_kwargs", "get_model_device", "compile_model", "is_compiled_model"]
