"""VR180 projection on the HIP engine.  Mirrors ``iw3/equirectangular.py`` ``equirectangular_projection`` :7-40: zero-pad to
1.5 x the long edge, then a bicubic ``grid_sample`` (zeros, align_corners) on the tan / cos grid, clamp — one kernel
(``nunif_hip_equirectangular``), the padded image never exists."""
from . import _ops


def equirectangular_projection(c, device=None):
    """CHW float -> C x Hp x Wp (``device`` is accepted for signature parity; the tensor must already be on the GPU)."""
    if device is not None:
        c = c.to(device)
    return _ops.equirectangular(c)
