"""Oracle: iw3 ``sbs.mlbw`` (multi-layer backward warp: ``--method mlbw_l2 / mlbw_l4 / mlbw_l2s / mlbw_l4s``), torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference: ``iw3/models/mlbw.py`` — ``WABlock`` :18-34, ``MLBW.__init__`` :42-76 (lv1_in 1x9 conv,
lv2 window-attention blocks with alternating zero-pad shift, lv1_out 1x9 conv), ``_calc_pad`` :78-93 (eval: centred
replicate padding to multiples of 32 x 4), ``_forward`` :95-116, ``_forward_delta_only`` :238-247;
``nunif/modules/attention.py`` ``WindowMHA2d`` :118-161 (shift = ZERO padding by half a window on both sides, then crop);
``iw3/backward_warp.py`` ``pad_delta_y`` :239-243, ``apply_divergence_nn_delta_weight`` :262-341 (layer weights resized
with bilinear + antialias, composite = sum_i backward_warp(c, delta_i) * w_i, clamp).
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
    delta, weight = x.chunk(2, dim=1)
    return delta, F.softmax(weight.float(), dim=1)


def apply_divergence_nn_delta_weight(sd, c, depth, divergence, convergence, shift, num_layers):
    if shift > 0:
        c, depth = torch.flip(c, (3,)), torch.flip(depth, (3,))
    B, _, H, W = depth.shape
    delta, weight = delta_forward(sd, RF.make_input(depth, divergence, convergence, max(H, W)), num_layers)
    if c.shape[2:] != weight.shape[2:]:
        weight = F.interpolate(weight, size=c.shape[-2:], mode="bilinear", align_corners=True, antialias=True)
    grid = RF.make_grid(B, W, H)
    scale = torch.tensor(1.0 / (W // 2 - 1))
    z = torch.zeros_like(c)
    for i in range(num_layers):
        d = torch.cat([delta[:, i:i + 1], torch.zeros_like(delta[:, i:i + 1])], dim=1)
        z = z + RF.backward_warp(c, grid, d, scale) * weight[:, i:i + 1]
    z = z.clamp(0, 1)
    return torch.flip(z, (3,)) if shift > 0 else z


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
