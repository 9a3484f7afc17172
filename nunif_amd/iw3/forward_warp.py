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


def nonwarp_mask(c, depth, divergence, convergence, view="right"):
    """Pixels of ``c`` that a ``view`` eye synthesised from the OTHER eye could not see (reference :259-295): warp the depth
    to the other eye (filled), warp a dummy image back, and return that warp's hole mask.  Two ``nunif_hip_forward_warp`` calls;
    the left-view branch's flips are the reference's."""
    divergence = divergence * 0.5        # cancels the 2x multiplier of synthetic_view = right | left
    if c.shape[2] != depth.shape[2] or c.shape[3] != depth.shape[3]:
        depth = _ops.resize_aa(depth, c.shape[-2:], mode="bilinear", align_corners=True)
    depth3 = depth.repeat(1, 3, 1, 1)
    if view == "right":
        warped_depth, _ = depth_order_bilinear_forward_warp(depth3, depth, divergence, convergence, synthetic_view="left",
                                                            fill=True, return_mask=False)
        warped_depth = warped_depth.mean(dim=1, keepdim=True)
        _, _, _, mask = depth_order_bilinear_forward_warp(c.new_zeros(c.shape), warped_depth, divergence, convergence,
                                                          synthetic_view="right", fill=False, return_mask=True)
    else:
        c, depth, depth3 = c.flip(-1), depth.flip(-1), depth3.flip(-1)
        _, warped_depth = depth_order_bilinear_forward_warp(depth3, depth, divergence, convergence, synthetic_view="right",
                                                            fill=True, return_mask=False)
        warped_depth = warped_depth.mean(dim=1, keepdim=True)
        _, _, mask, _ = depth_order_bilinear_forward_warp(c.new_zeros(c.shape), warped_depth, divergence, convergence,
                                                          synthetic_view="left", fill=False, return_mask=True)
        c, mask = c.flip(-1), mask.flip(-1)
    return c, mask
