"""Oracle: iw3 ``sbs.row_flow_v3`` (the default ``--method``) and the NN backward-warp glue, torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference: ``iw3/models/row_flow_v3.py`` — ``WABlock`` :14-30, ``RowFlowV3._forward`` :56-66,
``_forward_delta_only`` :102-107; ``nunif/modules/attention.py`` — ``WindowMHA2d`` :118-161, ``MHA`` :94-115,
``sliced_sdp`` :61-77, ``WindowScoreBias`` :375-419 (+ ``_gen_window_score_bias_input`` :347-372);
``nunif/modules/permute.py`` — ``pixel_unshuffle`` :45-62, ``pixel_shuffle`` :65-82, ``bchw_to_bnc`` :85-103,
``bnc_to_bchw`` :112-128; ``nunif/modules/replication_pad2d.py`` ``replication_pad2d_naive`` :30-61;
``iw3/backward_warp.py`` — ``make_divergence_feature_value`` :8-14, ``make_input_tensor`` :17-64, ``backward_warp``
:67-83, ``make_grid`` :86-93, ``apply_divergence_nn_LR`` :124-160, ``apply_divergence_nn_delta`` :191-236.
State-dict keys are the reference's (``blocks.{0,1,2}…``, ``last_layer.1``).
"""
import math

import torch
import torch.nn.functional as F


# ---- WindowScoreBias ------------------------------------------------------------------------------------------------
def window_score_bias_input(window):
    """(index [N*N], unique normalised deltas [U,2]) exactly as _gen_window_score_bias_input builds them."""
    sh, sw = window
    pos = [(y, x) for y in range(sh) for x in range(sw)]
    delta = [(a[0] - b[0], a[1] - b[1]) for a in pos for b in pos]
    uniq = sorted(set(delta))
    index = torch.tensor([uniq.index(d) for d in delta], dtype=torch.int64)
    ud = torch.tensor(uniq, dtype=torch.float32)
    return index, ud / ud.abs().max()


def window_score_bias(sd, p, window):
    """[N,N] additive attention bias: to_bias MLP (Linear 2->h, GELU(erf), Linear h->1) on the relative offsets."""
    index, delta = window_score_bias_input(window)
    h = F.gelu(F.linear(delta, sd[p + "to_bias.0.weight"], sd[p + "to_bias.0.bias"]))
    b = F.linear(h, sd[p + "to_bias.2.weight"], sd[p + "to_bias.2.bias"])
    n = window[0] * window[1]
    return b[index].reshape(n, n)


# ---- window MHA -----------------------------------------------------------------------------------------------------------
def window_mha(sd, p, x, window, bias, num_heads=2):
    B, C, H, W = x.shape
    sh, sw = window
    oh, ow = H // sh, W // sw
    t = x.reshape(B, C, oh, sh, ow, sw).permute(0, 2, 4, 3, 5, 1).reshape(B * oh * ow, sh * sw, C)      # bchw_to_bnc
    qkv = F.linear(t, sd[p + "mha.qkv_proj.weight"], sd[p + "mha.qkv_proj.bias"])
    q, k, v = qkv.split(C, dim=-1)
    hd = C // num_heads
    n = sh * sw

    def heads(z):
        return z.view(-1, n, num_heads, hd).permute(0, 2, 1, 3)
    q, k, v = heads(q), heads(k), heads(v)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd)) + bias                     # SDPA with a float attn_mask
    o = torch.softmax(s, dim=-1) @ v
    o = o.permute(0, 2, 1, 3).reshape(-1, n, C)
    o = F.linear(o, sd[p + "mha.head_proj.weight"], sd[p + "mha.head_proj.bias"])
    return o.reshape(B, oh, ow, sh, sw, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W)                 # bnc_to_bchw


def wa_block(sd, p, x, window):
    x = x + window_mha(sd, p + "mha.", x, window, window_score_bias(sd, p + "bias.", window))
    z = F.gelu(F.conv2d(x, sd[p + "conv_mlp.0.weight"], sd[p + "conv_mlp.0.bias"]))
    z = F.conv2d(F.pad(z, (1, 1, 1, 1), mode="replicate"), sd[p + "conv_mlp.3.weight"], sd[p + "conv_mlp.3.bias"])
    return x + F.leaky_relu(z, 0.1)


def pixel_unshuffle_w(x, sw):
    B, C, H, W = x.shape
    return x.reshape(B, C, H, 1, W // sw, sw).permute(0, 1, 3, 5, 2, 4).reshape(B, C * sw, H, W // sw)


def pixel_shuffle_w(x, sw):
    B, C, H, W = x.shape
    return x.reshape(B, C // sw, 1, sw, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, C // sw, H, W * sw)


def delta_forward(sd, x):
    """RowFlowV3._forward: x [B,3,h,w] (depth, divergence feature, convergence feature) -> delta [B,1,h,w]."""
    h, w = x.shape[2:]
    pad1 = 96 - w % 96            # (mod = 12) * 8; always >= 1 column / row of padding (row_flow_v3.py:58-59)
    pad2 = 12 - h % 12
    x = F.pad(x, (0, pad1, 0, pad2), mode="replicate")
    x = pixel_unshuffle_w(x, 8)
    x = F.conv2d(x, sd["blocks.0.weight"], sd["blocks.0.bias"])
    x = wa_block(sd, "blocks.1.", x, (4, 4))
    x = wa_block(sd, "blocks.2.", x, (3, 3))
    x = pixel_shuffle_w(x, 8)
    x = x[:, :, :h, :w]
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd["last_layer.1.weight"], sd["last_layer.1.bias"])


# ---- NN backward-warp glue ---------------------------------------------------------------------------------------------------
def divergence_feature_values(divergence, convergence, image_width):
    pix = divergence * 0.5 * 0.01 * image_width
    return pix / 32.0, (-pix * convergence) / 32.0


def make_input(depth, divergence, convergence, image_width):
    """depth [B,1,h,w] -> [B,3,h,w]  (make_input_tensor with c=None, no mapper, no screen-border taper)."""
    dv, cv = divergence_feature_values(divergence, convergence, image_width)
    return torch.cat([depth, torch.full_like(depth, dv), torch.full_like(depth, cv)], dim=1)


def make_grid(B, W, H):
    my, mx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    return torch.stack([mx, my])[None].expand(B, 2, H, W)


def backward_warp(c, grid, delta, delta_scale):
    grid = grid + delta * delta_scale
    if c.shape[2:] != grid.shape[2:]:
        grid = F.interpolate(grid, size=c.shape[-2:], mode="bilinear", align_corners=True, antialias=False)
    z = F.grid_sample(c, grid.permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=True)
    return torch.clamp(z, 0, 1)


def apply_divergence_nn_delta(sd, c, depth, divergence, convergence, shift, steps=1):
    """One eye (apply_divergence_nn_delta :191-236): the right eye runs on horizontally flipped inputs.  ``steps`` > 1: the
    net runs ``steps`` times at divergence / steps on the depth warped by the previous flows (:209-222), then the image
    is warped by the flows one after the other (:224-227)."""
    if shift > 0:
        c, depth = torch.flip(c, (3,)), torch.flip(depth, (3,))
    B, _, H, W = depth.shape
    grid, scale = make_grid(B, W, H), torch.tensor(1.0 / (W // 2 - 1))
    depth_warp, deltas = depth, []
    for j in range(steps):
        delta = delta_forward(sd, make_input(depth_warp, divergence / steps, convergence, max(H, W)))
        delta = torch.cat([delta, torch.zeros_like(delta)], dim=1)
        deltas.append(delta)
        if j + 1 < steps:
            depth_warp = backward_warp(depth_warp, grid, delta, scale)
    z = c
    for delta in deltas:
        z = backward_warp(z, grid, delta, scale)
    return torch.flip(z, (3,)) if shift > 0 else z


def apply_divergence_nn_LR(sd, c, depth, divergence, convergence, synthetic_view="both", steps=1):
    if synthetic_view == "both":
        return (apply_divergence_nn_delta(sd, c, depth, divergence, convergence, -1, steps),
                apply_divergence_nn_delta(sd, c, depth, divergence, convergence, 1, steps))
    if synthetic_view == "right":
        return c, apply_divergence_nn_delta(sd, c, depth, divergence * 2, convergence, 1, steps)
    return apply_divergence_nn_delta(sd, c, depth, divergence * 2, convergence, -1, steps), c


def apply_divergence_nn_symmetric(sd, c, depth, divergence, convergence, synthetic_view="both"):
    """iw3/backward_warp.py:343-379 (model.symmetric): one un-flipped flow, +delta / -delta, no clamp beyond backward_warp's."""
    B, _, H, W = depth.shape
    if synthetic_view != "both":
        divergence = divergence * 2
    delta = delta_forward(sd, make_input(depth, divergence, convergence, W))
    delta = torch.cat([delta, torch.zeros_like(delta)], dim=1)
    grid, scale = make_grid(B, W, H), 1.0 / (W // 2 - 1)
    left = backward_warp(c, grid, delta, scale) if synthetic_view != "right" else c
    right = backward_warp(c, grid, -delta, scale) if synthetic_view != "left" else c
    return left, right


def random_state_dict(*args, **kwargs):
    """Seeded test weights: alias of ``nunif_amd.synthetic.row_flow_v3_state_dict`` (moved there so that bench.py and the tools do
    not import the oracle for their inputs)."""
    from nunif_amd.synthetic import row_flow_v3_state_dict
    return row_flow_v3_state_dict(*args, **kwargs)
