"""Backward warps on the HIP engine.  Mirrors ``iw3/backward_warp.py``: ``apply_divergence_grid_sample`` :96-121
(``make_grid`` :86-93 and ``backward_warp`` :67-83 folded into ``nunif_hip_backward_warp``) and the NN-delta path the
default ``--method row_flow_v3`` takes — ``make_divergence_feature_value`` :8-14, ``make_input_tensor`` :17-64 (c=None),
``apply_divergence_nn_LR`` :124-160, ``apply_divergence_nn`` :163-188, ``apply_divergence_nn_delta`` :191-236.
and the multi-layer variant ``apply_divergence_nn_delta_weight`` :262-341 (MLBW).  ``apply_divergence_nn_symmetric`` :343-379, the hole-mask output (``postprocess_hole_mask`` :382-393, ``nonwarp_mask``
:396-422).  The cycle (training) output is not provided.
"""
import torch

from . import _ops
from .mapper import get_mapper


def apply_divergence_grid_sample(c, depth, divergence, convergence, synthetic_view):
    assert synthetic_view in {"both", "right", "left"}
    left, right = _ops.backward_warp(c, depth, divergence, convergence, synthetic_view)
    return (c if left is None else left.to(c.dtype)), (c if right is None else right.to(c.dtype))


def make_grid(batch, width, height, device):
    """The identity sampling grid of ``backward_warp`` (reference :86-93): x, y planes of linspace(-1, 1).  The HIP warps build
    this grid on the fly, so the tensor only carries a marker that ``backward_warp`` recognises."""
    mesh_y, mesh_x = torch.meshgrid(torch.linspace(-1, 1, height, device=device), torch.linspace(-1, 1, width, device=device),
                                    indexing="ij")
    grid = torch.cat((mesh_x.reshape(1, 1, height, width).expand(batch, 1, height, width),
                      mesh_y.reshape(1, 1, height, width).expand(batch, 1, height, width)), dim=1)
    grid._nunif_identity_grid = True
    return grid


def pad_delta_y(delta_x):
    """[B,C,H,W] x-displacements -> [B,2C,H,W] with zero y planes interleaved (reference :239-243)."""
    B, C, H, W = delta_x.shape
    return torch.stack([delta_x, torch.zeros_like(delta_x)], dim=2).reshape(B, C * 2, H, W)


def backward_warp(c, grid, delta, delta_scale):
    """``clamp(grid_sample(c, grid + delta * delta_scale, bilinear, border, align_corners=True), 0, 1)`` with the grid resized
    bilinearly (align_corners) to the image when the delta map is smaller (reference :67-83).  ``grid`` must come from
    ``make_grid`` and ``delta`` from ``pad_delta_y`` (x displacement, zero y) — what every call site in the reference passes;
    one launch of ``nunif_hip_delta_warp``."""
    if not getattr(grid, "_nunif_identity_grid", False):
        raise NotImplementedError("backward_warp on the HIP engine takes the identity grid of make_grid()")
    if delta.shape[1] != 2:
        raise ValueError("delta must be [B,2,h,w] (x, y) as built by pad_delta_y")
    return _ops.delta_warp(c, delta[:, 0:1].contiguous(), delta_scale).to(c.dtype)


def make_divergence_feature_value(divergence, convergence, image_width):
    divergence_pix = divergence * 0.5 * 0.01 * image_width
    divergence_feature_value = divergence_pix / 32.0
    convergence_feature_value = (-divergence_pix * convergence) / 32.0
    return divergence_feature_value, convergence_feature_value


def make_input_tensor(c, depth, divergence, convergence, image_width, mapper=None, preserve_screen_border=False):
    """CHW depth -> the 3-plane feature tensor (depth | divergence | convergence).  ``c`` must be None (the inference
    form); the screen-border taper multiplies the two constant planes by linear ramps exactly like the reference."""
    if c is not None:
        raise NotImplementedError("make_input_tensor with an image (training layout) is not used at inference")
    depth = depth.squeeze(0)
    if mapper is not None:
        depth = get_mapper(mapper)(depth)
    divergence_value, convergence_value = make_divergence_feature_value(divergence, convergence, image_width)
    divergence_feat = torch.full_like(depth, divergence_value)
    if torch.is_tensor(convergence_value):
        convergence_feat = convergence_value.to(depth.device).expand_as(depth).clone()
    else:
        convergence_feat = torch.full_like(depth, convergence_value)
    if preserve_screen_border:
        border_pix = round(divergence * 0.75 * 0.01 * image_width * (depth.shape[-1] / image_width))
        if border_pix > 0:
            wl = torch.linspace(0.0, 1.0, border_pix, device=depth.device)[None, :]
            wr = torch.linspace(1.0, 0.0, border_pix, device=depth.device)[None, :]
            for feat in (divergence_feat, convergence_feat):
                feat[:, :border_pix] = wl * feat[:, :border_pix]
                feat[:, -border_pix:] = wr * feat[:, -border_pix:]
    return torch.stack([depth, divergence_feat, convergence_feat], dim=0)


def make_input_batch(depth, divergence, convergence, image_width, preserve_screen_border=False):
    """``torch.stack([make_input_tensor(None, depth[i], ...) for i in range(B)])`` (reference :206-211 / :280-285).  With a
    scalar convergence on a device batch this is ONE launch of ``nunif_hip_make_input_planes`` (the reference's per-item
    full_like / linspace / slice-multiply / stack chain is ~10 ATen kernels per frame and eye); per-item convergence tensors
    take the per-item form."""
    B = depth.shape[0]
    if depth.is_cuda and not torch.is_tensor(convergence) and not (isinstance(convergence, (list, tuple))
                                                                   and len(set(map(float, convergence))) > 1):
        conv = float(convergence[0]) if isinstance(convergence, (list, tuple)) else float(convergence)
        dv, cv = make_divergence_feature_value(divergence, conv, image_width)
        border_pix = 0
        if preserve_screen_border:
            border_pix = max(round(divergence * 0.75 * 0.01 * image_width * (depth.shape[-1] / image_width)), 0)
        return _ops.make_input_planes(depth, dv, cv, border_pix)
    conv = convergence.flatten() if torch.is_tensor(convergence) else (
        list(convergence) if isinstance(convergence, (list, tuple)) else [convergence] * B)
    return torch.stack([make_input_tensor(None, depth[i], divergence=divergence, convergence=conv[i], image_width=image_width,
                                          preserve_screen_border=preserve_screen_border) for i in range(B)])


def apply_divergence_nn_delta(model, c, depth, divergence, convergence, steps, shift, preserve_screen_border=False,
                              enable_amp=True):
    """One eye.  The reference flips ``c`` and ``depth`` for the right eye, runs the same net and flips the result back;
    here the mirror is folded into the first and the last kernel (no flipped copies)."""
    assert model.delta_output
    steps = 1 if steps is None else int(steps)
    flip = shift > 0
    B, _, H, W = depth.shape
    base_size = max(H, W)
    if torch.is_tensor(convergence):
        convergence = convergence.flatten()
    else:
        convergence = [convergence] * B
    # (the screen-border taper is mirror-symmetric — linspace(0,1,n) on the left, linspace(1,0,n) on the right — so the
    #  planes can be built un-mirrored and mirrored together with the depth inside the first kernel)
    delta_scale = 1.0 / (W // 2 - 1)
    divergence_step = divergence / steps
    # warp_steps > 1 (reference :190-231; calc_auto_warp_steps turns it on for row_flow_v3 at divergence > 5): every step
    # runs the net on the depth warped by the previous steps' flows at divergence / steps, then the image is warped by
    # the flows one after the other (each warp clamps to [0, 1], :81).  The mirror of the right eye stays folded into the
    # kernels: both the net and the warp read mirrored and write un-mirrored, so every intermediate — the warped depth of
    # step j included — lives in un-mirrored coordinates and the same ``flip`` goes to every call.
    depth_warp = depth
    deltas = []
    for j in range(steps):
        x = make_input_batch(depth_warp, divergence_step, convergence, base_size, preserve_screen_border)
        delta = model.infer_delta(x, flip=flip)
        deltas.append(delta)
        if j + 1 < steps:
            depth_warp = _ops.delta_warp(depth_warp, delta, delta_scale, flip=flip)
    z = c
    for delta in deltas:
        z = _ops.delta_warp(z, delta, delta_scale, flip=flip)
    return z.to(c.dtype)


def apply_divergence_nn_delta_weight(model, c, depth, divergence, convergence, steps, shift, preserve_screen_border=False,
                                     enable_amp=True, return_mask=False):
    """MLBW: L flows + softmax layer weights; composite = clamp(sum_i backward_warp(c, delta_i) * w_i)."""
    assert model.delta_output
    flip = shift > 0
    hole_mask = bool(getattr(model, "hole_mask", False))
    B, _, H, W = depth.shape
    base_size = max(H, W)
    if torch.is_tensor(convergence):
        convergence = convergence.flatten()
    else:
        convergence = [convergence] * B
    x = make_input_batch(depth, divergence, convergence, base_size, preserve_screen_border)
    if hole_mask:
        delta, layer_weight, hole_mask_logits = model.infer_delta(x, flip=flip)   # logits already in image coordinates
    else:
        delta, layer_weight = model.infer_delta(x, flip=flip)
        hole_mask_logits = None
    if c.shape[2] != layer_weight.shape[2] or c.shape[3] != layer_weight.shape[3]:
        # F.interpolate(layer_weight, size, bilinear, align_corners=True, antialias=True)  :295-297
        layer_weight = _ops.resize_aa(layer_weight, c.shape[-2:], mode="bilinear", align_corners=True)
    delta_scale = 1.0 / (W // 2 - 1)
    z = _ops.delta_weight_warp(c, delta, layer_weight, delta_scale, flip=flip)
    if return_mask:
        return z.to(c.dtype), hole_mask_logits
    if hole_mask_logits is not None:
        # "hole fill for visualize" :333-339: z = z * (1 - mask), fused into the mask kernel
        _ops.hole_mask_postprocess(hole_mask_logits, c.shape[-2:], 0.15, z=z)
    return z.to(c.dtype)


def _scaled_iter(n_iter, width, base_width):
    """dilate_inner / dilate_outer (iw3/dilation.py:74-103): n_iter <= 0 is a no-op, else max(round(W / base * n), 1)."""
    if n_iter <= 0:
        return 0
    return max(round(width / base_width * n_iter), 1) if base_width is not None else n_iter


def postprocess_hole_mask(mask_logits, target_size, threshold, inner_dilation=0, outer_dilation=0):
    """Reference :382-393: closing(n_iter=1) -> bilinear resize (align_corners) -> sigmoid > threshold ->
    dilate_inner / dilate_outer (horizontal OR-dilations scaled by target width / logit width).  Returns bool [B,1,H,W]."""
    base_width = mask_logits.shape[-1]
    width = int(target_size[1])
    return _ops.hole_mask_postprocess(mask_logits, target_size, threshold,
                                      inner_iter=_scaled_iter(inner_dilation, width, base_width),
                                      outer_iter=_scaled_iter(outer_dilation, width, base_width))


def nonwarp_mask(model, c, depth, divergence, convergence, mapper=None, threshold=0.15, inner_dilation=0, outer_dilation=0):
    """Reference :396-422: warp the depth to the left, warp it back to the right with the hole-mask model and return
    ``(c, mask)`` — the disocclusion mask of the un-warped view."""
    disparity = get_mapper(mapper)(depth) if mapper is not None else depth
    warped_depth, _ = apply_divergence_nn_delta_weight(model, depth, disparity, divergence=divergence,
                                                       convergence=convergence, steps=1, shift=-1,
                                                       preserve_screen_border=False, enable_amp=True, return_mask=True)
    disparity = get_mapper(mapper)(warped_depth) if mapper is not None else depth
    dummy = torch.zeros_like(c)
    _, mask_logits = apply_divergence_nn_delta_weight(model, dummy, disparity, divergence=divergence,
                                                      convergence=convergence, steps=1, shift=1,
                                                      preserve_screen_border=False, enable_amp=True, return_mask=True)
    mask = postprocess_hole_mask(mask_logits, c.shape[-2:], threshold=threshold, inner_dilation=inner_dilation,
                                 outer_dilation=outer_dilation)
    return c, mask


def apply_divergence_nn_symmetric(model, c, depth, divergence, convergence, synthetic_view, enable_amp):
    """Reference :343-379 (``row_flow_v3_sym``): ONE flow from the un-flipped planes (feature width = W, not max(H, W));
    the left eye samples at grid + delta, the right eye at grid - delta."""
    assert synthetic_view in {"both", "right", "left"}
    assert model.delta_output and model.symmetric
    B, _, H, W = depth.shape
    if synthetic_view != "both":
        divergence = divergence * 2
    if torch.is_tensor(convergence):
        convergence = convergence.flatten()
    else:
        convergence = [convergence] * B
    x = torch.stack([make_input_tensor(None, depth[i], divergence=divergence, convergence=convergence[i], image_width=W)
                     for i in range(B)])
    delta = model.infer_delta(x, flip=False)
    delta_scale = 1.0 / (W // 2 - 1)
    left_eye = _ops.delta_warp(c, delta, delta_scale, flip=False).to(c.dtype) if synthetic_view != "right" else c
    right_eye = _ops.delta_warp(c, delta, -delta_scale, flip=False).to(c.dtype) if synthetic_view != "left" else c
    return left_eye, right_eye


def apply_divergence_nn(model, c, depth, divergence, convergence, steps, shift, preserve_screen_border=False,
                        enable_amp=True):
    if model.name == "sbs.mlbw":
        return apply_divergence_nn_delta_weight(model, c, depth, divergence=divergence, convergence=convergence,
                                                steps=steps, shift=shift, preserve_screen_border=preserve_screen_border,
                                                enable_amp=enable_amp)
    return apply_divergence_nn_delta(model, c, depth, divergence=divergence, convergence=convergence, steps=steps,
                                     shift=shift, preserve_screen_border=preserve_screen_border, enable_amp=enable_amp)


def apply_divergence_nn_LR(model, c, depth, divergence, convergence, steps, synthetic_view="both",
                           preserve_screen_border=False, enable_amp=True):
    assert synthetic_view in {"both", "right", "left"}
    steps = 1 if steps is None else steps
    if getattr(model, "symmetric", False):
        return apply_divergence_nn_symmetric(model, c, depth, divergence, convergence, synthetic_view=synthetic_view,
                                             enable_amp=enable_amp)
    kw = dict(preserve_screen_border=preserve_screen_border, enable_amp=enable_amp)
    if synthetic_view == "both":
        left_eye = apply_divergence_nn(model, c, depth, divergence, convergence, steps, shift=-1, **kw)
        right_eye = apply_divergence_nn(model, c, depth, divergence, convergence, steps, shift=1, **kw)
    elif synthetic_view == "right":
        left_eye = c
        right_eye = apply_divergence_nn(model, c, depth, divergence * 2, convergence, steps, shift=1, **kw)
    else:
        left_eye = apply_divergence_nn(model, c, depth, divergence * 2, convergence, steps, shift=-1, **kw)
        right_eye = c
    return left_eye, right_eye
