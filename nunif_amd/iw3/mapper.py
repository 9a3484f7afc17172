"""Depth -> disparity mappers on the HIP engine.  Mirrors ``iw3/mapper.py`` (reference): ``resolve_mapper_function``
:64-118, ``get_mapper`` :129-151 (":" chains and "a+b=w" blends), ``get_mapper_levels`` :167-186,
``resolve_mapper_name`` :195-232.  Each elementary mapper is one ``nunif_hip_map_depth`` launch."""
import math

from . import _ops

_MUL = {"mul_1": (0.343, 12), "mul_2": (0.515, 12), "mul_3": (0.687, 12)}
_INV_MUL = {"inv_mul_1": (-0.002102, 7.8788), "inv_mul_2": (-0.0003, 6.2626), "inv_mul_3": (-0.0001, 3.4343)}
_SHIFT = {"shift_30": 3.0, "shift_20": 2.0, "shift_14": 1.4, "shift_08": 0.8, "shift_06": 0.6, "shift_045": 0.45}
_DIV = {"div_25": 2.5, "div_10": 1, "div_6": 0.6, "div_4": 0.4, "div_2": 0.2, "div_1": 0.1}


# named elementary mappers (reference :7-61), each one launch of nunif_hip_map_depth
def softplus01_legacy(depth, c=6):
    return _ops.map_depth(depth, 2, c)


def softplus01(x, bias, scale):
    return _ops.map_depth(x, 3, bias, scale)


def inv_softplus01(x, bias, scale):
    return _ops.map_depth(x, 4, bias, scale)


def distance_to_disparity(x, c):
    return _ops.map_depth(x, 5, c)


def shift_relative_depth(x, min_distance, max_distance=16):
    return _ops.map_depth(x, 6, min_distance, max_distance)


def resolve_mapper_function(name):
    if name == "none":
        return lambda x: x
    if name == "pow2":
        return lambda x: _ops.map_depth(x, 1)
    if name == "softplus":
        return lambda x: _ops.map_depth(x, 2, 6)
    if name == "softplus2":
        return lambda x: _ops.map_depth(_ops.map_depth(x, 2, 6), 1)
    if name in _MUL:
        return lambda x: _ops.map_depth(x, 3, *_MUL[name])
    if name in _INV_MUL:
        return lambda x: _ops.map_depth(x, 4, *_INV_MUL[name])
    if name in _SHIFT:
        return lambda x: _ops.map_depth(x, 6, _SHIFT[name], 16)
    if name in _DIV:
        return lambda x: _ops.map_depth(x, 5, _DIV[name])
    raise NotImplementedError(f"mapper={name}")


def chain(x, functions):
    """iw3/mapper.py:123-126."""
    for f in functions:
        x = f(x)
    return x


def get_mapper(name):
    functions = []
    for part in name.split(":"):
        if "+" in part:
            pair, weight = part.split("=")
            weight = 0.5 if not weight else float(weight)
            assert 0.0 <= weight <= 1.0
            a, b = (resolve_mapper_function(n) for n in pair.split("+"))
            functions.append(lambda x, a=a, b=b, w=weight: a(x) * (1 - w) + b(x) * w)
        else:
            functions.append(resolve_mapper_function(part))

    def run(x):
        for f in functions:
            x = f(x)
        return x
    return run


METRIC_DIV_MAPPER = ["none", "div_25", "div_10", "div_6", "div_4", "div_2", "div_1"]
RELATIVE_MUL_MAPPER = ["inv_mul_3", "inv_mul_2", "inv_mul_1", "none", "mul_1", "mul_2", "mul_3"]
RELATIVE_SHIFT_MAPPER = ["shift_045", "shift_06", "shift_08", "none", "shift_14", "shift_20", "shift_30"]


def get_mapper_levels(metric_depth, mapper_type=None):
    if metric_depth:
        if mapper_type in (None, "div"):
            return METRIC_DIV_MAPPER
        raise ValueError(f"{mapper_type} is not metric depth mapper")
    if mapper_type in (None, "mul"):
        return RELATIVE_MUL_MAPPER
    if mapper_type == "shift":
        return RELATIVE_SHIFT_MAPPER
    raise ValueError(f"{mapper_type} is not relative depth mapper")


def resolve_mapper_name(mapper, foreground_scale, metric_depth, mapper_type=None):
    if mapper is not None:
        if mapper == "auto":
            return "div_6" if metric_depth else "none"
        return mapper
    levels = get_mapper_levels(metric_depth=metric_depth, mapper_type=mapper_type)
    if float(foreground_scale).is_integer():
        return levels[int(foreground_scale) + 3]
    sign = 1 if foreground_scale > 0 else -1
    mag = abs(foreground_scale)
    lo, hi = math.floor(mag), math.ceil(mag)
    return f"{levels[sign * lo + 3]}+{levels[sign * hi + 3]}={round(mag - lo, 2)}"
