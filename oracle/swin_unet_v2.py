"""Oracle: waifu2x ``swin_unet_v2`` family (``waifu2x.swin_unet_v2_1x / _2x / _4x / _1xs``), torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The HIP engine does not carry this family yet: this file and
``tests/golden/swin_unet_v2.npz`` pin the reference behaviour the kernels will be held to.

Follows ``waifu2x/models/swin_unet_v2.py`` (reference): ``GLUConvMLP`` :14-35, ``MLP`` :53-68, ``WACBlock`` :71-101 (window MHA with a
learned score bias and a bias-free LayerNorm applied to the windowed tokens AFTER the zero-pad shift, then a GLU 1x1 -> replicate-pad
3x3 conv MLP, LeakyReLU 0.2), ``IR`` :132-155, ``PatchDown`` :158-183 (2x2 s2 conv + pixel-unshuffle channel-mean shortcut),
``PatchUp`` :186-209 (1x1 conv + pixel shuffle + channel-repeat shortcut), ``ToImage`` :212-226, ``SourceResidual`` :229-272,
``get_shift_config`` :275-282, ``SwinUNetV2Base._forward`` :337-352; ``nunif/modules/attention.py`` ``WindowMHA2d`` :118-161,
``MHA`` :94-115, ``WindowScoreBias`` :375-419."""
import math

import torch
import torch.nn.functional as F

from . import row_flow_v3 as RF


def window_mha(sd, p, x, ws, bias, num_heads, shift, norm_weight):
    pad = ws // 2 if shift else 0
    if pad:
        x = F.pad(x, (pad, pad, pad, pad), mode="constant", value=0)
    B, C, H, W = x.shape
    oh, ow = H // ws, W // ws
    t = x.reshape(B, C, oh, ws, ow, ws).permute(0, 2, 4, 3, 5, 1).reshape(B * oh * ow, ws * ws, C)       # bchw_to_bnc
    t = F.layer_norm(t, (C,), norm_weight, None, 1e-5)
    q, k, v = F.linear(t, sd[p + "mha.qkv_proj.weight"], sd[p + "mha.qkv_proj.bias"]).split(C, dim=-1)
    hd, n = C // num_heads, ws * ws

    def heads(z):
        return z.view(-1, n, num_heads, hd).permute(0, 2, 1, 3)
    s = (heads(q) @ heads(k).transpose(-1, -2)) * (1.0 / math.sqrt(hd)) + bias
    o = (torch.softmax(s, dim=-1) @ heads(v)).permute(0, 2, 1, 3).reshape(-1, n, C)
    o = F.linear(o, sd[p + "mha.head_proj.weight"], sd[p + "mha.head_proj.bias"])
    o = o.reshape(B, oh, ow, ws, ws, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W)                    # bnc_to_bchw
    return o[:, :, pad:H - pad, pad:W - pad] if pad else o


def wac_block(sd, p, x, num_heads, ws, shift, conv_mlp=True):
    bias = RF.window_score_bias(sd, p + "relative_bias.", (ws, ws))
    x = x + window_mha(sd, p + "mha.", x, ws, bias, num_heads, shift, sd[p + "norm.weight"])
    z = F.conv2d(x, sd[p + "conv_mlp.w1.weight"], sd[p + "conv_mlp.w1.bias"])
    if conv_mlp:                                                   # GLUConvMLP
        z = F.glu(z, dim=1)
        z = F.conv2d(F.pad(z, (1, 1, 1, 1), mode="replicate"), sd[p + "conv_mlp.w2.weight"], sd[p + "conv_mlp.w2.bias"])
        return x + F.leaky_relu(z, 0.2)
    z = F.conv2d(F.leaky_relu(z, 0.1), sd[p + "conv_mlp.w2.weight"], sd[p + "conv_mlp.w2.bias"])          # MLP
    return x + z


def shift_config(n):
    return tuple(reversed([i % 2 == 1 for i in range(n)]))


def n_blocks(sd, p):
    return sum(1 for k in sd if k.startswith(p) and k.endswith("mha.mha.qkv_proj.weight"))


def model_forward(sd, x, scale, raw=False):
    """SwinUNet{1,2,4}xV2.forward in eval mode: x [B,3,T,T] -> clamp([B,3,T*s - 2*offset, ...])."""
    u = "unet."
    src = x
    x1 = F.leaky_relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd[u + "ir.path1.0.weight"], sd[u + "ir.path1.0.bias"]), 0.2)
    x2 = F.conv2d(F.pixel_unshuffle(x, 2), sd[u + "ir.path2.1.weight"], sd[u + "ir.path2.1.bias"])
    x2 = wac_block(sd, u + "ir.path2.2.", x2, 2, 8, True)
    x2 = wac_block(sd, u + "ir.path2.3.", x2, 2, 8, False)
    x = torch.cat([x1, F.pixel_shuffle(x2, 2)], dim=1)
    x = F.conv2d(x, sd[u + "patch.weight"], sd[u + "patch.bias"])
    x = F.leaky_relu(x[:, :, 7:-7, 7:-7], 0.2)
    C = x.shape[1]
    C2 = sd[u + "down1.conv.weight"].shape[0]
    heads, heads2 = max(C // 32, 2), max(C2 // 32, 2)
    n1, n3 = n_blocks(sd, u + "wac1."), n_blocks(sd, u + "wac3.")
    windows1 = [8, 6][:n1]
    for i, sh in enumerate(shift_config(n1)):
        x = wac_block(sd, f"{u}wac1.blocks.{i}.", x, heads, windows1[i], sh)
    skip = x
    sc = F.pixel_unshuffle(x, 2)
    B, C4, H, W = sc.shape
    sc = sc.view(B, C2, C4 // C2, H, W).mean(dim=2)
    x = sc + F.leaky_relu(F.conv2d(x, sd[u + "down1.conv.weight"], sd[u + "down1.conv.bias"], stride=2), 0.2)
    for i, sh in enumerate(shift_config(4)):
        x = wac_block(sd, f"{u}wac2.blocks.{i}.", x, heads2, 8, sh)
    sc = F.pixel_shuffle(x.repeat_interleave(C * 4 // C2, dim=1), 2)
    x = sc + F.pixel_shuffle(F.leaky_relu(F.conv2d(x, sd[u + "up1.proj.weight"], sd[u + "up1.proj.bias"]), 0.2), 2)
    x = x + skip
    for i, sh in enumerate(shift_config(n3)):
        x = wac_block(sd, f"{u}wac3.blocks.{i}.", x, heads, 8, sh, conv_mlp=(i < n3 - 1))
    x = F.conv2d(x, sd[u + "to_residual_image.proj.weight"], sd[u + "to_residual_image.proj.bias"])
    if scale > 1:
        x = F.pixel_shuffle(x, scale)
    x = x[:, :, scale:-scale, scale:-scale]
    s = F.conv2d(F.pad(src, (1, 1, 1, 1), mode="replicate"), sd[u + "to_image.resampling.weight"])
    if scale > 1:
        s = F.pixel_shuffle(s, scale)
    unpad = (s.shape[2] - x.shape[2]) // 2
    if unpad:
        s = s[:, :, unpad:-unpad, unpad:-unpad]
    z = s + x * sd[u + "to_image.scale_bias"]
    return z if raw else torch.clamp(z, 0.0, 1.0)


def randomize(sd, seed):
    """Re-draw every bias / norm weight / the scale_bias of a freshly constructed reference model (its defaults are zeros / ones,
    which would hide bias-handling bugs) and damp the residual branches like ``nunif_amd.synthetic.swin_unet_state_dict`` does."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if k.endswith(".index") or k.endswith(".delta"):
            pass
        elif k.endswith("norm.weight"):
            v = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("scale_bias"):
            v = torch.full(v.shape, 0.06)          # keeps src + scale_bias * residual inside [0, 1] for these weights
        elif k.endswith(".bias"):
            v = 0.02 * torch.randn(v.shape, generator=g)
        elif k.endswith("head_proj.weight") or k.endswith("conv_mlp.w2.weight"):
            v = 0.8 * v
        out[k] = v
    return out
