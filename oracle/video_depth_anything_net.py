"""Oracle: the Video-Depth-Anything network behind ``infer_video_depth_one`` / ``reset_state`` (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED.**

The reference does not contain this network: ``iw3/video_depth_anything_streaming_model.py:58-65`` loads it with
``torch.hub.load("nagadomi/Video-Depth-Anything_iw3:main", "VideoDepthAnythingStreaming", encoder=...)`` and uses
``model.infer_video_depth_one(frame, use_amp)`` (:94), ``model.reset_state()`` (:75), ``model.head`` / ``model.pretrained``
(:139-141).  Neither that repository nor its weights exist offline and no second implementation is installed here
(``transformers`` has Depth-Anything, not Video-Depth-Anything), so nothing below has ever been compared with the real thing.
It restates the PUBLISHED architecture (Video-Depth-Anything, ``video_depth_anything/video_depth.py``, ``dpt_temporal.py``,
``motion_module/motion_module.py``), from memory of the public code:

* encoder: DINOv2 ViT (``pretrained.*``), taps 2-5-8-11 (vits / vitb) or 4-11-17-23 (vitl) through the final norm — the same
  encoder as Depth-Anything V2 (``oracle/depth_anything_v2.encoder_features``, which IS pinned against HuggingFace);
* ``head.*`` = the DPT head of Depth-Anything V2 plus four ``motion_modules``: on ``layer_3`` and ``layer_4`` (after
  ``resize_layers``, before ``layer{3,4}_rn``), on ``path_4`` (the output of ``refinenet4``) and on ``path_3``;
* a motion module (AnimateDiff's ``TemporalModule``): GroupNorm(32, eps 1e-6) -> ``proj_in`` Linear -> ONE transformer block
  (two temporal self-attention blocks, each ``x + to_out(attn(LayerNorm(x)))`` with 8 heads over the FRAME axis at a fixed
  pixel, sinusoidal position encoding of the frame index added to the attention's input, ``to_q / to_k / to_v`` without bias;
  then ``x + FF(LayerNorm(x))`` with a GEGLU feed-forward of inner width 4 C) -> ``proj_out`` Linear -> + the module's input.

What the STREAMING form does with its caches is the part restated with the least certainty; the policy here (and in the HIP
engine, ``nunif_amd/csrc/depth_temporal.hip``) is: every temporal attention block keeps the LayerNorm'ed hidden states of the
previous frames (at most ``MAX_LEN - 1 = 31``); a new frame attends to [cache, itself] with the position encoding applied to the
index INSIDE that window (so a cached frame's encoding moves as the window slides), then joins the cache and the oldest entry
leaves.  The first frame after ``reset_state`` attends to itself only.  ``forward_window`` is the published offline ``forward``
(all T <= 32 frames attend to each other); with T = 1 and an empty cache the two forms coincide (tests).
"""
import math

import torch
import torch.nn.functional as F

from . import depth_anything_v2 as DA

MAX_LEN = 32            # temporal_max_len = num_frames
HEADS = 8               # num_attention_heads
GN_GROUPS = 32
PATCH = DA.PATCH


def positional_encoding(d_model, max_len=MAX_LEN):
    """AnimateDiff ``PositionalEncoding``: pe[pos, 2i] = sin(pos w_i), pe[pos, 2i + 1] = cos(pos w_i), w_i = 10000^(-2i / d)."""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def new_state():
    """What ``reset_state`` leaves: no cached frames in any of the 4 x 2 temporal attention blocks."""
    return {"caches": [[None, None] for _ in range(4)], "frames": 0}


def temporal_module(sd, p, x, cache=None):
    """x: [T, C, H, W], the T frames of ONE video at this stage.  cache None: offline, the T frames attend to each other.
    cache = [c0, c1] (each None or [L, H*W, C]): streaming, T == 1.  -> (y [T, C, H, W], new cache)."""
    T, C, H, W = x.shape
    t = p + "temporal_transformer."
    h = F.group_norm(x, GN_GROUPS, sd[t + "norm.weight"], sd[t + "norm.bias"], eps=1e-6)
    h = h.permute(0, 2, 3, 1).reshape(T, H * W, C)
    h = F.linear(h, sd[t + "proj_in.weight"], sd[t + "proj_in.bias"])
    b = t + "transformer_blocks.0."
    hd = C // HEADS
    new_cache = []
    for a in range(2):
        n = F.layer_norm(h, (C,), sd[f"{b}norms.{a}.weight"], sd[f"{b}norms.{a}.bias"])          # eps 1e-5 (nn.LayerNorm default)
        seq = n if cache is None or cache[a] is None else torch.cat([cache[a], n], dim=0)
        assert seq.shape[0] <= MAX_LEN
        new_cache.append(seq[-(MAX_LEN - 1):])
        L = seq.shape[0]
        ab = f"{b}attention_blocks.{a}."
        pe = sd.get(ab + "pos_encoder.pe")
        pe = positional_encoding(C)[:L] if pe is None else pe.reshape(-1, C)[:L]
        kv_in = seq + pe[:, None, :]
        q = F.linear(kv_in[-T:], sd[ab + "to_q.weight"]).reshape(T, H * W, HEADS, hd).permute(1, 2, 0, 3)       # [P, heads, T, hd]
        k = F.linear(kv_in, sd[ab + "to_k.weight"]).reshape(L, H * W, HEADS, hd).permute(1, 2, 0, 3)
        v = F.linear(kv_in, sd[ab + "to_v.weight"]).reshape(L, H * W, HEADS, hd).permute(1, 2, 0, 3)
        att = torch.softmax((q * hd ** -0.5) @ k.transpose(-2, -1), dim=-1) @ v
        att = att.permute(2, 0, 1, 3).reshape(T, H * W, C)
        h = h + F.linear(att, sd[ab + "to_out.0.weight"], sd[ab + "to_out.0.bias"])
    n = F.layer_norm(h, (C,), sd[b + "ff_norm.weight"], sd[b + "ff_norm.bias"])
    g = F.linear(n, sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"])
    val, gate = g.chunk(2, dim=-1)                                                                       # diffusers GEGLU
    h = h + F.linear(val * F.gelu(gate), sd[b + "ff.net.2.weight"], sd[b + "ff.net.2.bias"])
    h = F.linear(h, sd[t + "proj_out.weight"], sd[t + "proj_out.bias"])
    return h.reshape(T, H, W, C).permute(0, 3, 1, 2) + x, new_cache


def head(sd, feats, gh, gw, caches=None):
    """``DPTHeadTemporal.forward`` on the T frames in ``feats`` (4 x [T, gh*gw, embed]) -> ([T, 1, 14 gh, 14 gw], new caches)."""
    p = "head."
    raw = []
    for i, f in enumerate(feats):
        x = f.permute(0, 2, 1).reshape(f.shape[0], f.shape[2], gh, gw)
        x = F.conv2d(x, sd[f"{p}projects.{i}.weight"], sd[f"{p}projects.{i}.bias"])
        if i == 0:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.0.weight"], sd[p + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.1.weight"], sd[p + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = F.conv2d(x, sd[p + "resize_layers.3.weight"], sd[p + "resize_layers.3.bias"], stride=2, padding=1)
        raw.append(x)
    c = caches if caches is not None else [None] * 4
    new = [None] * 4
    raw[2], new[0] = temporal_module(sd, p + "motion_modules.0.", raw[2], c[0])
    raw[3], new[1] = temporal_module(sd, p + "motion_modules.1.", raw[3], c[1])
    l1, l2, l3, l4 = (F.conv2d(x, sd[f"{p}scratch.layer{i + 1}_rn.weight"], None, padding=1) for i, x in enumerate(raw))
    s = p + "scratch."
    path4 = DA._fusion(sd, s + "refinenet4.", l4, size=l3.shape[2:])
    path4, new[2] = temporal_module(sd, p + "motion_modules.2.", path4, c[2])
    path3 = DA._fusion(sd, s + "refinenet3.", path4, l3, size=l2.shape[2:])
    path3, new[3] = temporal_module(sd, p + "motion_modules.3.", path3, c[3])
    path2 = DA._fusion(sd, s + "refinenet2.", path3, l2, size=l1.shape[2:])
    path1 = DA._fusion(sd, s + "refinenet1.", path2, l1)
    out = F.conv2d(path1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], padding=1)
    out = F.interpolate(out, (gh * PATCH, gw * PATCH), mode="bilinear", align_corners=True)
    out = F.relu(F.conv2d(out, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], padding=1))
    out = F.relu(F.conv2d(out, sd[s + "output_conv2.2.weight"], sd[s + "output_conv2.2.bias"]))
    return out, new


def forward_window(sd, frames, taps=None):
    """The published offline ``forward`` on ONE clip: frames [T, 3, h, w] (T <= 32) -> [T, h, w]."""
    feats, gh, gw = DA.encoder_features(sd, frames, taps)
    out, _ = head(sd, feats, gh, gw, None)
    return F.relu(out).squeeze(1)


def infer_video_depth_one(sd, frame, state, taps=None):
    """The streaming step: frame [3, h, w] ImageNet-normalised, h and w multiples of 14; ``state`` from ``new_state()`` is
    updated in place.  -> [1, h, w]."""
    feats, gh, gw = DA.encoder_features(sd, frame.unsqueeze(0), taps)
    out, state["caches"] = head(sd, feats, gh, gw, state["caches"])
    state["frames"] += 1
    return F.relu(out).squeeze(1)


def random_state_dict(*args, **kwargs):
    """Seeded test weights: alias of ``nunif_amd.synthetic.video_depth_anything_state_dict``."""
    from nunif_amd.synthetic import video_depth_anything_state_dict
    return video_depth_anything_state_dict(*args, **kwargs)
