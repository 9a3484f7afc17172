"""Oracle: iw3 ``iw3.depth_aa`` (depth anti-aliasing net, ``--depth-aa``), torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference ``iw3/models/depth_aa.py``: ``WABlock`` :11-26 (WindowMHA2d 2 heads, window 8, zero-pad shift of
4; conv_mlp 1x1 / GELU / replicate-pad 3x3 / LeakyReLU(0.1); WindowScoreBias(8)), ``DepthAA.forward`` :59-85 (centred
replicate pad to multiples of 16 — always at least 1 px —, pixel_unshuffle 2, proj_in 4->32, three blocks with shift
True / False / True, proj_out 32->4, pixel_shuffle 2, crop, residual, optional clamp) and ``DepthAA.infer`` :46-56
(tensor-wide min-max normalise -> forward(clamp=False) -> de-normalise).
"""
import math

import torch
import torch.nn.functional as F

from . import row_flow_v3 as RF


def window_mha8(sd, p, x, bias, shift):
    if shift:
        x = F.pad(x, (4, 4, 4, 4), mode="constant", value=0)
    x = RF.window_mha(sd, p, x, (8, 8), bias, num_heads=2)
    return x[:, :, 4:-4, 4:-4] if shift else x


def wa_block(sd, p, x, shift):
    x = x + window_mha8(sd, p + "mha.", x, RF.window_score_bias(sd, p + "bias.", (8, 8)), shift)
    z = F.gelu(F.conv2d(x, sd[p + "conv_mlp.0.weight"], sd[p + "conv_mlp.0.bias"]))
    z = F.conv2d(F.pad(z, (1, 1, 1, 1), mode="replicate"), sd[p + "conv_mlp.3.weight"], sd[p + "conv_mlp.3.bias"])
    return x + F.leaky_relu(z, 0.1)


def forward(sd, x, clamp=True):
    src = x
    h, w = x.shape[2:]
    pad_w, pad_h = 16 - w % 16, 16 - h % 16
    pw1, ph1 = pad_w // 2, pad_h // 2
    pw2, ph2 = pad_w - pw1, pad_h - ph1
    x = F.pad(x, (pw1, pw2, ph1, ph2), mode="replicate")
    x = F.pixel_unshuffle(x, 2)
    x = F.conv2d(x, sd["proj_in.weight"], sd["proj_in.bias"])
    for i, shift in enumerate((True, False, True)):
        x = wa_block(sd, f"blocks.{i}.", x, shift)
    x = F.conv2d(x, sd["proj_out.weight"], sd["proj_out.bias"])
    x = F.pixel_shuffle(x, 2)
    x = x[:, :, ph1:x.shape[2] - ph2, pw1:x.shape[3] - pw2]
    x = src + x
    return torch.clamp(x, 0, 1) if clamp else x


def infer(sd, x):
    mn, mx = x.amin(), x.amax()
    scale = mx - mn
    z = torch.nan_to_num((x - mn) / scale)
    return forward(sd, z, clamp=False) * scale + mn


def random_state_dict(seed):
    """Seeded weights in the reference's key layout, all biases non-zero; proj_out (zero-initialised in the reference
    constructor) is given small weights so that the net really changes the depth (a few percent of its range)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, *shape, std=None, bstd=0.05):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        sd[key + ".bias"] = rnd(shape[0], std=bstd)

    lin("proj_in", 32, 4, 1, 1, std=0.7)
    for i in range(3):
        p = f"blocks.{i}."
        lin(p + "mha.mha.qkv_proj", 96, 32)
        lin(p + "mha.mha.head_proj", 32, 32, std=0.5 * math.sqrt(1.0 / 32))
        lin(p + "conv_mlp.0", 32, 32, 1, 1)
        lin(p + "conv_mlp.3", 32, 32, 3, 3, std=0.5 * math.sqrt(1.0 / 288))
        lin(p + "bias.to_bias.0", 16, 2, std=1.0, bstd=0.3)
        lin(p + "bias.to_bias.2", 1, 16, std=1.0, bstd=0.3)
        sd[p + "bias.index"], sd[p + "bias.delta"] = RF.window_score_bias_input((8, 8))
    lin("proj_out", 4, 32, 1, 1, std=0.006, bstd=0.003)
    return sd
