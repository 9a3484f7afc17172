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


def random_state_dict(seed, num_layers=2, small=False):
    """Seeded weights in the reference's key layout (every bias non-zero); the output conv is scaled so that the layer
    deltas differ by a few depth pixels and the layer-weight logits really select between them."""
    g = torch.Generator().manual_seed(seed)
    C = 32 * num_layers
    sd = {}

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, *shape, std=None, bstd=0.05):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        sd[key + ".bias"] = rnd(shape[0], std=bstd)

    lin("lv1_in.1", C // 8, 3, 1, 9, std=math.sqrt(2.0 / 27))
    for i in range(2 if small else 4):
        p = f"lv2.{i}."
        lin(p + "mha.mha.qkv_proj", 3 * C, C)
        lin(p + "mha.mha.head_proj", C, C, std=0.5 * math.sqrt(1.0 / C))
        lin(p + "conv_mlp.0", C, C, 1, 1)
        lin(p + "conv_mlp.3", C, C, 3, 3, std=0.5 * math.sqrt(1.0 / (9 * C)))
        lin(p + "bias.to_bias.0", 8, 2, std=1.0, bstd=0.3)
        lin(p + "bias.to_bias.2", 1, 8, std=1.0, bstd=0.3)
        sd[p + "bias.index"], sd[p + "bias.delta"] = RF.window_score_bias_input((4, 4))
    lin("lv1_out.1", 2 * num_layers, C // 8, 1, 9, std=2.0 * math.sqrt(1.0 / (9 * C // 8)), bstd=1.0)
    return sd
