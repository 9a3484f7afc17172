"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

An emulation of the reference's OWN GPU arithmetic — ``torch.autocast(device_type="cuda", dtype=torch.float16)``
(``nunif/device.py:58-71``, entered by ``Waifu2x.render`` / ``BaseDepthModel.infer`` whenever a CUDA device is used) — on the CPU
fp32 oracle: inside ``with fp16_autocast_emulation():`` the result of EVERY torch operation on fp32 tensors is rounded to the
nearest fp16 value (and kept in an fp32 container).  ``F.linear`` / ``conv2d`` / ``matmul`` then behave like their fp16 tensor-core
forms (fp16 operands, fp32 accumulation, fp16 result), elementwise ops and residual adds like fp16 kernels, and ops autocast runs in
fp32 (softmax, layer_norm) are computed in fp32 from fp16-valued inputs and rounded where their single consumer (a matmul /
Linear) would cast them anyway.  It rounds in a few places where autocast keeps fp32 for one more op (scores + fp32 bias table), so
it is a slight LOWER bound of the reference's fp16 accuracy.

Used by the trained-regime parity tests (``tests/test_gpu_hot_regime.py``): with weights in the regime of a trained net (large
relative-position tables, outlier channels, undamped residual branches) the distance between ANY fp16 engine and the fp32 oracle
is dominated by fp16 storage itself, so the HIP engine is held to "no worse than the reference's own fp16 mode" instead of to a
fixed 50 dB.
"""
import torch
from torch.overrides import TorchFunctionMode


def half_weights(sd):
    """autocast casts the PARAMETERS of every Linear / conv to fp16 as well (they are leaves, no op inside the mode produces
    them): the state dict rounded the same way, for use inside ``fp16_autocast_emulation``.  Tables that autocast leaves in
    fp32 (relative-position bias: added to the scores after the matmul) keep their precision."""
    keep = ("relative_position_bias_table", "relative_position_index", "norm")
    return {k: (v.half().float() if v.is_floating_point() and not any(t in k for t in keep) else v) for k, v in sd.items()}


def _round(t):
    if isinstance(t, torch.Tensor) and t.dtype == torch.float32:
        return t.half().float()
    return t


class fp16_autocast_emulation(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = getattr(func, "__name__", "")
        if name.endswith("_") and not name.endswith("__") and isinstance(out, torch.Tensor) and out.dtype == torch.float32:
            with torch._C.DisableTorchFunction():
                out.copy_(out.half().float())                  # in-place op: keep the aliasing, round the storage
            return out
        if isinstance(out, torch.Tensor):
            with torch._C.DisableTorchFunction():
                return _round(out)
        if isinstance(out, (tuple, list)):
            with torch._C.DisableTorchFunction():
                return type(out)(_round(o) for o in out)
        return out
