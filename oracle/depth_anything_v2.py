"""Oracle: Depth-Anything V1 / V2 / V2-metric (DINOv2 ViT-S / B / L /14 encoder + DPT head), torch CPU fp32 — pinned against
HuggingFace ``transformers`` (an independent implementation), not against the hub repository the reference loads.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference does not contain this network: ``iw3/depth_anything_model.py:200-230`` loads it with
``torch.hub.load("nagadomi/Depth-Anything_iw3", "DepthAnything", encoder="v2_vits")`` (branch ``main``, unpinned), and
neither that repository nor its weights exist offline.  This file restates the PUBLISHED architecture
(Depth-Anything-V2 ``dpt.py`` / ``dinov2.py``: ViT-S/14, 12 blocks with LayerScale, features from blocks 2/5/8/11
through the final norm without the class token; DPT head with out_channels 48/96/192/384 and 64 fusion features) from
the call-site contract in SURVEY.md §8c: input B x 3 x h x w ImageNet-normalised with h, w multiples of 14, output
B x h x w (ReLU'd inverse depth).  State-dict key names follow the public checkpoint (``pretrained.*``, ``depth_head.*``)
so that a real ``depth_anything_v2_vits.pth`` can be tried as soon as one is available.  Pinning:
``tests/test_depth_anything_vs_hf.py`` loads the same weights into ``transformers.DepthAnythingForDepthEstimation``
(``oracle/hf_pin.py``; ViT-S / ViT-B, V1 taps, metric head: max |diff| <= 5e-5) and ``tests/golden/depth_anything_hf.npz``
holds HuggingFace outputs that the GPU test compares the HIP engine with.  One divergence between upstream and HuggingFace
exists — the position-embedding resize (``scale_factor=(g + 0.1) / 37`` upstream vs ``size=``) — this file follows upstream
(what the reference's hub fork wraps); see ``oracle/hf_pin.py``.  The hub fork itself has never been run against this.

The geometry is read from the state dict (``config_of``): embed 384 / 768 / 1024 with heads of 64, 12 / 24 blocks, DPT
out_channels (48-96-192-384 / 96-192-384-768 / 256-512-1024-1024) and fusion width (64 / 128 / 256) as published for vits /
vitb / vitl.  ``taps`` are the V2 ``intermediate_layer_idx`` (2-5-8-11 / 4-11-17-23); Depth-Anything V1 takes the last four
blocks (``get_intermediate_layers(x, 4)``).  ``max_depth`` > 0 is the V2 metric head: Sigmoid instead of the last ReLU and the
model output scaled by max_depth (hypersim 20, vkitti 80).
"""
import math

import torch
import torch.nn.functional as F

EMBED, HEADS, DEPTH, PATCH, MLP = 384, 6, 12, 14, 1536
TAPS = (2, 5, 8, 11)
OUT_CH = (48, 96, 192, 384)
FEAT = 64


def config_of(sd, taps=None):
    embed = sd["pretrained.patch_embed.proj.bias"].shape[0]
    depth = sum(1 for k in sd if k.startswith("pretrained.blocks.") and k.endswith(".attn.qkv.weight"))
    if taps is None:
        taps = {12: (2, 5, 8, 11), 24: (4, 11, 17, 23)}[depth]
    return {"embed": embed, "heads": embed // 64, "depth": depth, "taps": tuple(taps)}


def interpolate_pos_embed(pos_embed, gh, gw):
    """DINOv2 interpolate_pos_encoding for a gh x gw patch grid (bicubic, +0.1 offset, no antialias)."""
    EMBED = pos_embed.shape[-1]
    n = pos_embed.shape[1] - 1
    s = int(math.sqrt(n))
    if gh == s and gw == s:
        return pos_embed
    cls, patch = pos_embed[:, :1], pos_embed[:, 1:]
    patch = patch.reshape(1, s, s, EMBED).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, scale_factor=((gh + 0.1) / s, (gw + 0.1) / s), mode="bicubic", antialias=False)
    assert patch.shape[-2:] == (gh, gw)
    return torch.cat([cls, patch.permute(0, 2, 3, 1).reshape(1, gh * gw, EMBED)], dim=1)


def encoder_features(sd, x, taps=None):
    """-> 4 tensors [B, gh*gw, embed]: the tapped blocks through the final LayerNorm, class token dropped."""
    cfg = config_of(sd, taps)
    EMBED, HEADS, DEPTH, TAPS = cfg["embed"], cfg["heads"], cfg["depth"], cfg["taps"]
    B, _, h, w = x.shape
    gh, gw = h // PATCH, w // PATCH
    p = "pretrained."
    t = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=PATCH)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([sd[p + "cls_token"].expand(B, -1, -1), t], dim=1) + interpolate_pos_embed(sd[p + "pos_embed"], gh, gw)
    feats = []
    hd = EMBED // HEADS
    for i in range(DEPTH):
        b = f"{p}blocks.{i}."
        y = F.layer_norm(t, (EMBED,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], eps=1e-6)
        qkv = F.linear(y, sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"]).reshape(B, -1, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
        a = torch.softmax((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1), dim=-1) @ qkv[2]
        a = F.linear(a.transpose(1, 2).reshape(B, -1, EMBED), sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"])
        t = t + a * sd[b + "ls1.gamma"]
        y = F.layer_norm(t, (EMBED,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], eps=1e-6)
        y = F.linear(F.gelu(F.linear(y, sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"])), sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"])
        t = t + y * sd[b + "ls2.gamma"]
        if i in TAPS:
            feats.append(F.layer_norm(t, (EMBED,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)[:, 1:])
    return feats, gh, gw


def _rcu(sd, p, x):
    y = F.conv2d(F.relu(x), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    y = F.conv2d(F.relu(y), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    return y + x


def _fusion(sd, p, x, skip=None, size=None):
    if skip is not None:
        x = x + _rcu(sd, p + "resConfUnit1.", skip)
    x = _rcu(sd, p + "resConfUnit2.", x)
    if size is None:
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        x = F.interpolate(x, size=size, mode="bilinear", align_corners=True)
    return F.conv2d(x, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


def head(sd, feats, gh, gw, max_depth=0.0):
    p = "depth_head."
    layers = []
    for i, f in enumerate(feats):
        x = f.permute(0, 2, 1).reshape(f.shape[0], f.shape[2], gh, gw)
        x = F.conv2d(x, sd[f"{p}projects.{i}.weight"], sd[f"{p}projects.{i}.bias"])
        if i == 0:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.0.weight"], sd[p + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.1.weight"], sd[p + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = F.conv2d(x, sd[p + "resize_layers.3.weight"], sd[p + "resize_layers.3.bias"], stride=2, padding=1)
        layers.append(F.conv2d(x, sd[f"{p}scratch.layer{i + 1}_rn.weight"], None, padding=1))
    l1, l2, l3, l4 = layers
    s = p + "scratch."
    path4 = _fusion(sd, s + "refinenet4.", l4, size=l3.shape[2:])
    path3 = _fusion(sd, s + "refinenet3.", path4, l3, size=l2.shape[2:])
    path2 = _fusion(sd, s + "refinenet2.", path3, l2, size=l1.shape[2:])
    path1 = _fusion(sd, s + "refinenet1.", path2, l1)
    out = F.conv2d(path1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], padding=1)
    out = F.interpolate(out, (gh * PATCH, gw * PATCH), mode="bilinear", align_corners=True)
    out = F.relu(F.conv2d(out, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], padding=1))
    out = F.conv2d(out, sd[s + "output_conv2.2.weight"], sd[s + "output_conv2.2.bias"])
    return torch.sigmoid(out) if max_depth > 0 else F.relu(out)


def model_forward(sd, x, taps=None, max_depth=0.0):
    """x: [B,3,h,w] ImageNet-normalised, h and w multiples of 14 -> [B,h,w]: relu(inverse depth), larger = nearer — or, for the
    metric heads (max_depth > 0), sigmoid * max_depth = distance."""
    feats, gh, gw = encoder_features(sd, x, taps)
    out = head(sd, feats, gh, gw, max_depth)
    return (out * max_depth if max_depth > 0 else F.relu(out)).squeeze(1)


def random_state_dict(*args, **kwargs):
    """Seeded test weights: alias of ``nunif_amd.synthetic.depth_anything_v2_state_dict`` (moved there so that bench.py and the tools do
    not import the oracle for their inputs)."""
    from nunif_amd.synthetic import depth_anything_v2_state_dict
    return depth_anything_v2_state_dict(*args, **kwargs)
