"""Grid-sample backward warp on the HIP engine.  Mirrors ``iw3/backward_warp.py`` ``apply_divergence_grid_sample``
:96-121 (``make_grid`` :86-93 and ``backward_warp`` :67-83 are folded into ``nunif_hip_backward_warp``).

The NN-delta variants (``apply_divergence_nn_*``, row_flow / MLBW side models, :124-379) are "next" rows
 and are not provided here.
"""
from . import _ops


def apply_divergence_grid_sample(c, depth, divergence, convergence, synthetic_view):
    assert synthetic_view in {"both", "right", "left"}
    left, right = _ops.backward_warp(c, depth, divergence, convergence, synthetic_view)
    return (c if left is None else left.to(c.dtype)), (c if right is None else right.to(c.dtype))
