"""Oracle: iw3 ``sbs.mlbw`` (multi-layer backward warp: ``--method mlbw_l2 / mlbw_l4 / mlbw_l2s / mlbw_l4s``), torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference: ``iw3/models/mlbw.py`` — ``WABlock`` :18-34, ``MLBW.__init__`` :42-76 (lv1_in 1x9 conv,
lv2 window-attention blocks with alternating zero-pad shift, lv1_out 1x9 conv), ``_calc_pad`` :78-93 (eval: centred
replicate padding to multiples of 32 x 4), ``_forward`` :95-116, ``_forward_delta_only`` :238-247;
``nunif/modules/attention.py`` ``WindowMHA2d`` :118-161 (shift = ZERO padding by half a window on both sides, then crop);
``iw3/backward_warp.py`` ``pad_delta_y`` :239-243, ``apply_divergence_nn_delta_weight`` :262-341 (layer weights resized
with bilinear + antialias, composite = sum_i backward_warp(c, delta_i) * w_i, clamp; ``return_mask`` / hole fill :325-341),
``postprocess_hole_mask`` :382-393, ``nonwarp_mask`` :396-422; ``iw3/dilation.py`` ``dilate`` / ``erode`` :41-61,
``closing`` :64-71, ``dilate_outer`` :74-88, ``dilate_inner`` :91-103.  The hole-mask variant (``sbs.mask_mlbw_l2``
:275-278) is recognised by its 2L + 1 output channels.
"""
import math

import torch
import torch.nn.functional as F

from . import row_flow_v3 as RF


def window_mha_shift(sd, p, x, bias, num_heads, shift):
    """WindowMHA2d with window (4,4): zero-pad (not roll) by 2 where shift[i], attend, crop."""
    ph, pw = (2 if shift[0] else 0), (2 if shift[1] else 0)
    if ph or pw:
        x = F.pad(x, (pw, pw, ph, ph), mode="constant", value=0)
    x = RF.window_mha(sd, p, x, (4, 4), bias, num_heads=num_heads)
    if ph or pw:
        x = x[:, :, ph:x.shape[2] - ph, pw:x.shape[3] - pw]
    return x


def wa_block(sd, p, x, num_heads, shift):
    x = x + window_mha_shift(sd, p + "mha.", x, RF.window_score_bias(sd, p + "bias.", (4, 4)), num_heads, shift)
    z = F.gelu(F.conv2d(x, sd[p + "conv_mlp.0.weight"], sd[p + "conv_mlp.0.bias"]))
    z = F.conv2d(F.pad(z, (1, 1, 1, 1), mode="replicate"), sd[p + "conv_mlp.3.weight"], sd[p + "conv_mlp.3.bias"])
    return x + z                                                      # no activation after the 3x3 (mlbw.py:22-27)


def block_shifts(n_blocks):
    # full model: (T,T),(F,F),(T,T),(F,F); small: (F,T),(F,F)   (mlbw.py:57-68)
    return [(True, True), (False, False), (True, True), (False, False)] if n_blocks == 4 else [(False, True), (False, False)]


def delta_forward(sd, x, num_layers):
    """MLBW._forward (eval, no hole mask): x [B,3,h,w] -> delta [B,L,h,w], layer_weight [B,L,h,w] (softmax over L)."""
    h, w = x.shape[2:]
    pad_w, pad_h = 32 - w % 32, 4 - h % 4
    pw1, ph1 = pad_w // 2, pad_h // 2
    pw2, ph2 = pad_w - pw1, pad_h - ph1
    x = F.pad(x, (pw1, pw2, ph1, ph2), mode="replicate")
    x1 = F.leaky_relu(F.conv2d(F.pad(x, (4, 4, 0, 0), mode="replicate"), sd["lv1_in.1.weight"], sd["lv1_in.1.bias"]), 0.2)
    x = RF.pixel_unshuffle_w(x1, 8)
    n_blocks = sum(1 for k in sd if k.startswith("lv2.") and k.endswith("mha.mha.qkv_proj.weight"))
    for i, shift in enumerate(block_shifts(n_blocks)):
        x = wa_block(sd, f"lv2.{i}.", x, num_layers, shift)
    x = RF.pixel_shuffle_w(x, 8)
    x = F.conv2d(F.pad(x + x1, (4, 4, 0, 0), mode="replicate"), sd["lv1_out.1.weight"], sd["lv1_out.1.bias"])
    x = x[:, :, ph1:x.shape[2] - ph2, pw1:x.shape[3] - pw2]
    if x.shape[1] == 2 * num_layers + 1:                              # hole_mask=True (mlbw.py:104-106)
        delta, weight = x[:, :2 * num_layers].chunk(2, dim=1)
        return delta, F.softmax(weight.float(), dim=1), x[:, 2 * num_layers:].float()
    delta, weight = x.chunk(2, dim=1)
    return delta, F.softmax(weight.float(), dim=1)


# ---- hole mask ---------------------------------------------------------------------------------------------------------
def closing(mask, kernel_size=3, n_iter=2):
    mask = mask.float()
    pad = kernel_size // 2
    for _ in range(n_iter):
        mask = F.max_pool2d(mask, kernel_size=kernel_size, stride=1, padding=pad)
    for _ in range(n_iter):
        mask = -F.max_pool2d(-mask, kernel_size=kernel_size, stride=1, padding=pad)
    return mask


def _n_iter(n_iter, width, base_width):
    return max(round(width / base_width * n_iter), 1) if base_width is not None else n_iter


def dilate_outer(mask, n_iter, base_width=None):
    if n_iter <= 0:
        return mask
    mask = mask.bool()
    for _ in range(_n_iter(n_iter, mask.shape[-1], base_width)):
        mask = mask | F.pad(mask, (1, 0, 0, 0))[:, :, :, :-1]
    return mask


def dilate_inner(mask, n_iter, base_width=None):
    if n_iter <= 0:
        return mask
    mask = mask.bool()
    for _ in range(_n_iter(n_iter, mask.shape[-1], base_width)):
        mask = mask | F.pad(mask, (0, 1, 0, 0))[:, :, :, 1:]
    return mask


def postprocess_hole_mask(mask_logits, target_size, threshold, inner_dilation=0, outer_dilation=0):
    base_width = mask_logits.shape[-1]
    mask_logits = closing(mask_logits, n_iter=1)
    if tuple(target_size) != tuple(mask_logits.shape[-2:]):
        mask_logits = F.interpolate(mask_logits, size=tuple(target_size), mode="bilinear", align_corners=True, antialias=False)
    mask = torch.sigmoid(mask_logits) > threshold
    mask = dilate_inner(mask, inner_dilation, base_width)
    return dilate_outer(mask, outer_dilation, base_width)


def apply_divergence_nn_delta_weight(sd, c, depth, divergence, convergence, shift, num_layers, return_mask=False):
    if shift > 0:
        c, depth = torch.flip(c, (3,)), torch.flip(depth, (3,))
    B, _, H, W = depth.shape
    out = delta_forward(sd, RF.make_input(depth, divergence, convergence, max(H, W)), num_layers)
    delta, weight = out[:2]
    logits = out[2] if len(out) == 3 else None
    if c.shape[2:] != weight.shape[2:]:
        weight = F.interpolate(weight, size=c.shape[-2:], mode="bilinear", align_corners=True, antialias=True)
    grid = RF.make_grid(B, W, H)
    scale = torch.tensor(1.0 / (W // 2 - 1))
    z = torch.zeros_like(c)
    for i in range(num_layers):
        d = torch.cat([delta[:, i:i + 1], torch.zeros_like(delta[:, i:i + 1])], dim=1)
        z = z + RF.backward_warp(c, grid, d, scale) * weight[:, i:i + 1]
    z = z.clamp(0, 1)
    if shift > 0:
        z = torch.flip(z, (3,))
        logits = torch.flip(logits, (3,)) if logits is not None else None
    if return_mask:
        return z, logits
    if logits is not None:                                            # hole fill for visualize :333-339
        z = z * (1 - postprocess_hole_mask(logits, c.shape[-2:], 0.15).float())
    return z


def nonwarp_mask(sd, c, depth, divergence, convergence, num_layers, threshold=0.15, inner_dilation=0, outer_dilation=0):
    """backward_warp.py:396-422 with mapper=None."""
    warped_depth, _ = apply_divergence_nn_delta_weight(sd, depth, depth, divergence, convergence, -1, num_layers,
                                                       return_mask=True)
    _, logits = apply_divergence_nn_delta_weight(sd, torch.zeros_like(c), depth, divergence, convergence, 1, num_layers,
                                                 return_mask=True)
    return c, postprocess_hole_mask(logits, c.shape[-2:], threshold, inner_dilation, outer_dilation)


def apply_divergence_nn_LR(sd, c, depth, divergence, convergence, num_layers, synthetic_view="both"):
    f = lambda div, sh: apply_divergence_nn_delta_weight(sd, c, depth, div, convergence, sh, num_layers)   # noqa: E731
    if synthetic_view == "both":
        return f(divergence, -1), f(divergence, 1)
    if synthetic_view == "right":
        return c, f(divergence * 2, 1)
    return f(divergence * 2, -1), c


def random_state_dict(*args, **kwargs):
    """Seeded test weights: alias of ``nunif_amd.synthetic.mlbw_state_dict`` (moved there so that bench.py and the tools do
    not import the oracle for their inputs)."""
    from nunif_amd.synthetic import mlbw_state_dict
    return mlbw_state_dict(*args, **kwargs)
