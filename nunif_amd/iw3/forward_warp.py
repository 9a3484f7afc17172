"""Depth-ordered bilinear forward warp on the HIP engine.

Mirrors ``iw3/forward_warp.py`` (reference): ``apply_divergence_forward_warp`` :246-256 and
``depth_order_bilinear_forward_warp`` :140-243 — same signatures and return conventions.  The whole algorithm
(splat z-test, shift_fill, fix_layered_holes, masks, fill/clamp) is one row-local kernel,
``nunif_hip_forward_warp`` (nunif_amd/csrc/iw3_warp.hip); unlike the reference it does not touch the process-global
``torch.use_deterministic_algorithms`` switch (forward_warp.py:96-108), so no lock is needed around it.
"""
from . import _ops


def depth_order_bilinear_forward_warp(c, depth, divergence, convergence, fill=True, synthetic_view="both",
                                      return_mask=False, inconsistent_shift=False, width_base=True):
    assert synthetic_view in {"both", "right", "left"}
    if inconsistent_shift:
        raise NotImplementedError("inconsistent_shift=True has no closed form and is not supported by the HIP engine")
    src_image = c
    if c.shape[2] != depth.shape[2] or c.shape[3] != depth.shape[3]:
        # bilinear + align_corners=True + antialias=True (the hybrid mapping)
        depth = _ops.resize_aa(depth, c.shape[-2:], mode="bilinear", align_corners=True)
    left, right, lmask, rmask = _ops.forward_warp(c, depth, divergence, convergence, fill, synthetic_view,
                                                  return_mask, width_base)
    left = src_image if left is None else left.to(c.dtype)
    right = src_image if right is None else right.to(c.dtype)
    if return_mask:
        return left, right, lmask, rmask
    return left, right


def apply_divergence_forward_warp(c, depth, divergence, convergence, method=None, synthetic_view="both",
                                  return_mask=False, inconsistent_shift=False, width_base=True):
    return depth_order_bilinear_forward_warp(c, depth, divergence, convergence, fill=(method == "forward_fill"),
                                             synthetic_view=synthetic_view, return_mask=return_mask,
                                             inconsistent_shift=inconsistent_shift, width_base=width_base)
