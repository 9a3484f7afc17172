"""``--method forward_inpaint``: un-filled depth-ordered forward warp + hole masks, an inpaint net repairs the holes of each eye.
Reference: ``iw3/forward_inpaint.py`` — ``forward_right`` / ``forward_left`` :18-40, ``ForwardInpaintImage`` :43-103,
``ForwardInpaintVideo`` :106-232, ``ForwardInpaint`` :235-300.

The family is two functions handed to the shared driver (``side_model.py``): the warp (``nunif_hip_forward_warp`` with
``return_mask=True``: warp + masks in one row-local pass) and the mask rule (mask > 0, ``mask_closing``, ``dilate_outer``,
``dilate_inner`` — ``nunif_hip_mask_morphology``).  Models are passed in as objects (the reference downloads them by name)."""
from .dilation import dilate_inner, dilate_outer, mask_closing
from .forward_warp import apply_divergence_forward_warp
from .side_model import SideModel, SideModelSpec


def _hole_mask(mask, eye_hw, p):
    mask = mask_closing(mask > 0)
    mask = dilate_outer(mask, n_iter=p.outer_dilation, base_width=p.base_width)
    return dilate_inner(mask, n_iter=p.inner_dilation, base_width=p.base_width)


def _warp(x, depth, divergence, convergence, synthetic_view, **_):
    return apply_divergence_forward_warp(x, depth, divergence=divergence, convergence=convergence,
                                         synthetic_view=synthetic_view, return_mask=True, width_base=False)


SPEC = SideModelSpec(warp=_warp, hole_mask=_hole_mask, mask_at_depth_size=False)


class ForwardInpaint(SideModel):
    """``ForwardInpaint(image_net, video_model=None)``: a ``LightInpaintV1`` and, optionally, a ``LightVideoInpaintV1``."""

    def __init__(self, model, video_model=None):
        super().__init__(SPEC, model, video_model, what="ForwardInpaint(image_model, video_model)")
