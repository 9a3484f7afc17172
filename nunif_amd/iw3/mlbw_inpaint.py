"""``--method mlbw_l2_inpaint``: mask-MLBW backward warp of both eyes, hole logits -> hole masks, an inpaint net repairs the
holes.  Reference: ``iw3/mlbw_inpaint.py`` — ``apply_divergence`` :38-75, ``forward_right`` / ``forward_left`` :21-35,
``MLBWInpaintImage`` :78-157, ``MLBWInpaintVideo`` :160-293, ``MLBWInpaint`` :296-360.

Here the family is two functions handed to the shared driver (``side_model.py``): the warp (``sbs.mask_mlbw_l2`` through
``apply_divergence_nn_delta_weight(return_mask=True)``; a single synthesised eye takes twice the divergence) and the mask rule
(``postprocess_hole_mask``: closing, resize to the eye, sigmoid > 0.15, OR-dilations).  Models are passed in as objects (the
reference downloads them by name)."""
from .backward_warp import apply_divergence_nn_delta_weight, postprocess_hole_mask
from .side_model import SideModel, SideModelSpec

MASK_MLBW_THRESHOLD = 0.15


def _hole_mask(logits, eye_hw, p):
    return postprocess_hole_mask(logits, target_size=eye_hw, threshold=MASK_MLBW_THRESHOLD,
                                 inner_dilation=p.inner_dilation, outer_dilation=p.outer_dilation)


def _spec(mask_mlbw):
    mask_mlbw.delta_output = True

    def eye(c, depth, divergence, convergence, shift, preserve_screen_border, enable_amp):
        return apply_divergence_nn_delta_weight(mask_mlbw, c, depth, divergence=divergence, convergence=convergence, steps=1,
                                                shift=shift, preserve_screen_border=preserve_screen_border,
                                                enable_amp=enable_amp, return_mask=True)

    def warp(c, depth, divergence, convergence, synthetic_view, preserve_screen_border=False, enable_amp=True, **_):
        kw = dict(preserve_screen_border=preserve_screen_border, enable_amp=enable_amp)
        if synthetic_view == "both":
            (left, lmask), (right, rmask) = eye(c, depth, divergence, convergence, -1, **kw), eye(c, depth, divergence, convergence, 1, **kw)
        elif synthetic_view == "right":
            left, lmask = c, None
            right, rmask = eye(c, depth, divergence * 2, convergence, 1, **kw)
        else:
            left, lmask = eye(c, depth, divergence * 2, convergence, -1, **kw)
            right, rmask = c, None
        return left, right, lmask, rmask

    return SideModelSpec(warp=warp, hole_mask=_hole_mask, mask_at_depth_size=True)


def apply_divergence(model, c, depth, divergence, convergence, preserve_screen_border, synthetic_view, enable_amp):
    """Reference :38-75 — the warp on its own: ``(left_eye, right_eye, left_mask, right_mask)``."""
    return _spec(model).warp(c, depth, divergence, convergence, synthetic_view, preserve_screen_border=preserve_screen_border,
                             enable_amp=enable_amp)


class MLBWInpaint(SideModel):
    """``MLBWInpaint(image_net, mask_mlbw, video_model=None)``: ``image_net`` a ``LightInpaintV1``, ``mask_mlbw`` an
    ``MLBW(hole_mask=True)``, ``video_model`` a ``LightVideoInpaintV1`` (optional: without it only the image mode exists)."""

    def __init__(self, model, mask_mlbw, video_model=None):
        super().__init__(_spec(mask_mlbw), model, video_model, what="MLBWInpaint(image_model, mask_mlbw, video_model)")
        self.mask_mlbw = mask_mlbw
