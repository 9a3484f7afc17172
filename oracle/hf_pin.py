"""Independent implementations (HuggingFace ``transformers``) of the two EXTERNAL networks the reference pulls in, loaded
with the same weights as the oracle's restatements — the pin for ``oracle/tv_swin_block.py`` and
``oracle/depth_anything_v2.py``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by ``tests/`` and ``tests/golden/make_golden_hf.py``.

Why HuggingFace: neither torchvision (``waifu2x/models/swin_unet.py:9-12``) nor the ``torch.hub`` repository
``nagadomi/Depth-Anything_iw3`` (``iw3/depth_anything_model.py:200-230``) exists offline, but ``transformers`` does, and
its ``SwinLayer`` / ``Dinov2Backbone`` + ``DepthAnythingForDepthEstimation`` are separate code bases of the same published
networks whose converted checkpoints reproduce the originals.  Key maps below follow transformers'
``convert_depth_anything_to_hf.py`` / ``convert_dinov2_to_hf.py`` naming.

Divergences found and how they are handled (also DESIGN.md §2):

* **Position-embedding interpolation.**  Depth-Anything(-V2)'s ``dinov2.py`` resizes the 37 x 37 table with
  ``F.interpolate(scale_factor=((gh + 0.1) / 37, (gw + 0.1) / 37), mode="bicubic")`` (``interpolate_offset = 0.1``), i.e.
  the sampling step is 37 / (g + 0.1); HuggingFace's ``Dinov2Embeddings.interpolate_pos_encoding`` passes ``size=(gh, gw)``
  (step 37 / g).  ``DepthAnythingHF(upstream_pos_embed=True)`` (default) swaps in the upstream form — the restatement
  follows the upstream repository, which is what the reference's hub fork wraps; ``False`` keeps HuggingFace's, and
  ``tests/test_depth_anything_vs_hf.py`` records how far apart the two are on the final depth.
* **Final activation.**  Upstream V2 applies ReLU in the head and again in ``DepthAnythingV2.forward``; HuggingFace once.
  ReLU is idempotent: no numerical difference.
* **Head upsample size.**  Both use ``(gh * 14, gw * 14)``, bilinear, align_corners=True.  Same.
* **Swin: when the window covers an axis** torchvision disables the shift per axis, HuggingFace when min(H, W) <= window;
  waifu2x never reaches either (``swin_unet.py:183`` asserts multiples of the window, smallest map 12 x 12).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def hf_swin_layer(block_sd, dim, heads, shift, H, W, window=6, mlp_ratio=2.0, norm="none"):
    """HuggingFace ``SwinLayer`` carrying the weights of a torchvision-layout block state dict
    (``attn.qkv / attn.proj / attn.relative_position_bias_table / mlp.0 / mlp.3 / norm1 / norm2``)."""
    from transformers.models.swin import modeling_swin as hf
    cfg = hf.SwinConfig(window_size=window, mlp_ratio=mlp_ratio, qkv_bias=True, hidden_act="gelu", hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5)
    cfg._attn_implementation = "eager"
    layer = hf.SwinLayer(cfg, dim, (H, W), heads, drop_path_rate=0.0, shift_size=shift).eval()
    C = dim
    with torch.no_grad():
        att = layer.attention
        for i, p in enumerate((att.q_proj, att.k_proj, att.v_proj)):
            p.weight.copy_(block_sd["attn.qkv.weight"][i * C:(i + 1) * C])
            p.bias.copy_(block_sd["attn.qkv.bias"][i * C:(i + 1) * C])
        att.o_proj.weight.copy_(block_sd["attn.proj.weight"])
        att.o_proj.bias.copy_(block_sd["attn.proj.bias"])
        att.relative_position_bias.relative_position_bias_table.copy_(block_sd["attn.relative_position_bias_table"])
        layer.mlp.fc1.weight.copy_(block_sd["mlp.0.weight"])
        layer.mlp.fc1.bias.copy_(block_sd["mlp.0.bias"])
        layer.mlp.fc2.weight.copy_(block_sd["mlp.3.weight"])
        layer.mlp.fc2.bias.copy_(block_sd["mlp.3.bias"])
        if norm == "none":
            layer.layernorm_before = nn.Identity()
            layer.layernorm_after = nn.Identity()
        else:                                   # LayerNormNoBias (nunif/modules/norm.py:17-21)
            for mine, name in ((layer.layernorm_before, "norm1"), (layer.layernorm_after, "norm2")):
                mine.weight.copy_(block_sd[name + ".weight"])
                mine.bias.zero_()
    return layer


class HFSwinTransformerBlock(nn.Module):
    """torchvision's constructor signature and state-dict layout, HuggingFace's arithmetic: lets the REFERENCE's own
    ``SwinUNetBase`` (``waifu2x/models/swin_unet.py``) run over an implementation that is independent of
    ``oracle/tv_swin_block.py`` (``oracle.refstub.install(swin_block="hf")``)."""

    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, dropout=0.0, attention_dropout=0.0,
                 stochastic_depth_prob=0.0, norm_layer=nn.LayerNorm, attn_layer=None):
        super().__init__()
        from .tv_swin_block import SwinTransformerBlock as Holder
        # the holder only owns parameters with torchvision's names; its forward is never called
        holder = Holder(dim, num_heads, window_size, shift_size, mlp_ratio=mlp_ratio, norm_layer=norm_layer)
        self.norm1, self.attn, self.norm2, self.mlp = holder.norm1, holder.attn, holder.norm2, holder.mlp
        self._geom = (dim, num_heads, int(window_size[0]), int(shift_size[0]), float(mlp_ratio))
        assert window_size[0] == window_size[1] and shift_size[0] == shift_size[1]

    def forward(self, x):
        dim, heads, window, shift, mlp_ratio = self._geom
        B, H, W, C = x.shape
        norm = "none" if isinstance(self.norm1, nn.Identity) else "ln"
        layer = hf_swin_layer(self.state_dict(), dim, heads, shift, H, W, window, mlp_ratio, norm)
        return layer(x.reshape(B, H * W, C), (H, W))[0].reshape(B, H, W, C)


# --------------------------------------------------------------------------------------------------------------------
# Depth-Anything: checkpoint keys -> transformers.DepthAnythingForDepthEstimation


def depth_anything_hf_state_dict(sd, depth):
    """Public Depth-Anything(-V2) checkpoint keys (``pretrained.*`` / ``depth_head.*``) -> HuggingFace keys."""
    out = {}
    p = "pretrained."
    e = "backbone.embeddings."
    out[e + "cls_token"] = sd[p + "cls_token"]
    out[e + "position_embeddings"] = sd[p + "pos_embed"]
    out[e + "patch_embeddings.projection.weight"] = sd[p + "patch_embed.proj.weight"]
    out[e + "patch_embeddings.projection.bias"] = sd[p + "patch_embed.proj.bias"]
    for i in range(depth):
        b, h = f"{p}blocks.{i}.", f"backbone.encoder.layer.{i}."
        C = sd[b + "attn.qkv.weight"].shape[1]
        for j, n in enumerate(("query", "key", "value")):
            out[f"{h}attention.attention.{n}.weight"] = sd[b + "attn.qkv.weight"][j * C:(j + 1) * C]
            out[f"{h}attention.attention.{n}.bias"] = sd[b + "attn.qkv.bias"][j * C:(j + 1) * C]
        out[h + "attention.output.dense.weight"] = sd[b + "attn.proj.weight"]
        out[h + "attention.output.dense.bias"] = sd[b + "attn.proj.bias"]
        out[h + "layer_scale1.lambda1"] = sd[b + "ls1.gamma"]
        out[h + "layer_scale2.lambda1"] = sd[b + "ls2.gamma"]
        for n in ("norm1", "norm2"):
            out[h + n + ".weight"] = sd[b + n + ".weight"]
            out[h + n + ".bias"] = sd[b + n + ".bias"]
        for n in ("fc1", "fc2"):
            out[h + "mlp." + n + ".weight"] = sd[b + "mlp." + n + ".weight"]
            out[h + "mlp." + n + ".bias"] = sd[b + "mlp." + n + ".bias"]
    out["backbone.layernorm.weight"] = sd[p + "norm.weight"]
    out["backbone.layernorm.bias"] = sd[p + "norm.bias"]
    d = "depth_head."
    for i in range(4):
        r = f"neck.reassemble_stage.layers.{i}."
        out[r + "projection.weight"] = sd[f"{d}projects.{i}.weight"]
        out[r + "projection.bias"] = sd[f"{d}projects.{i}.bias"]
        if i != 2:
            out[r + "resize.weight"] = sd[f"{d}resize_layers.{i}.weight"]
            out[r + "resize.bias"] = sd[f"{d}resize_layers.{i}.bias"]
        out[f"neck.convs.{i}.weight"] = sd[f"{d}scratch.layer{i + 1}_rn.weight"]
    for k in (1, 2, 3, 4):                      # refinenet4 runs first = fusion_stage.layers.0
        src, dst = f"{d}scratch.refinenet{k}.", f"neck.fusion_stage.layers.{4 - k}."
        out[dst + "projection.weight"] = sd[src + "out_conv.weight"]
        out[dst + "projection.bias"] = sd[src + "out_conv.bias"]
        for u in (1, 2):
            for c in (1, 2):
                for t in ("weight", "bias"):
                    out[f"{dst}residual_layer{u}.convolution{c}.{t}"] = sd[f"{src}resConfUnit{u}.conv{c}.{t}"]
    for hfk, k in (("head.conv1", "output_conv1"), ("head.conv2", "output_conv2.0"), ("head.conv3", "output_conv2.2")):
        out[hfk + ".weight"] = sd[f"{d}scratch.{k}.weight"]
        out[hfk + ".bias"] = sd[f"{d}scratch.{k}.bias"]
    return out


def _upstream_pos_embed(self, embeddings, height, width):
    """Depth-Anything-V2 ``dinov2.py`` ``interpolate_pos_encoding`` (offset 0.1, scale_factor form), bound onto HuggingFace's
    ``Dinov2Embeddings`` in place of its ``size=`` form."""
    n = self.position_embeddings.shape[1] - 1
    s = int(round(n ** 0.5))
    gh, gw = height // self.patch_size, width // self.patch_size
    if gh == s and gw == s:
        return self.position_embeddings
    dim = embeddings.shape[-1]
    cls, patch = self.position_embeddings[:, :1], self.position_embeddings[:, 1:]
    patch = patch.reshape(1, s, s, dim).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, scale_factor=((gh + 0.1) / s, (gw + 0.1) / s), mode="bicubic", antialias=False)
    assert patch.shape[-2:] == (gh, gw)
    return torch.cat([cls, patch.permute(0, 2, 3, 1).reshape(1, gh * gw, dim)], dim=1)


def depth_anything_hf(sd, taps=None, max_depth=0.0, upstream_pos_embed=True):
    """-> ``transformers.DepthAnythingForDepthEstimation`` (eval, fp32, eager attention) with the weights of ``sd``."""
    import types
    from transformers import DepthAnythingConfig, DepthAnythingForDepthEstimation, Dinov2Config
    embed = sd["pretrained.patch_embed.proj.bias"].shape[0]
    depth = sum(1 for k in sd if k.startswith("pretrained.blocks.") and k.endswith(".attn.qkv.weight"))
    if taps is None:
        taps = {12: (2, 5, 8, 11), 24: (4, 11, 17, 23)}[depth]
    grid = int(round((sd["pretrained.pos_embed"].shape[1] - 1) ** 0.5))
    neck = [sd[f"depth_head.projects.{i}.weight"].shape[0] for i in range(4)]
    fusion = sd["depth_head.scratch.layer1_rn.weight"].shape[0]
    bcfg = Dinov2Config(hidden_size=embed, num_hidden_layers=depth, num_attention_heads=embed // 64, mlp_ratio=4,
                        image_size=grid * 14, patch_size=14, layer_norm_eps=1e-6, layerscale_value=1.0,
                        use_swiglu_ffn=False, hidden_act="gelu", qkv_bias=True, apply_layernorm=True,
                        reshape_hidden_states=False, out_indices=[t + 1 for t in taps], use_mask_token=False)
    cfg = DepthAnythingConfig(backbone_config=bcfg, patch_size=14, reassemble_hidden_size=embed,
                              reassemble_factors=[4, 2, 1, 0.5], neck_hidden_sizes=neck, fusion_hidden_size=fusion,
                              head_in_index=-1, head_hidden_size=32,
                              depth_estimation_type="metric" if max_depth > 0 else "relative",
                              max_depth=int(max_depth) if max_depth > 0 else 1)
    cfg._attn_implementation = "eager"
    bcfg._attn_implementation = "eager"
    model = DepthAnythingForDepthEstimation(cfg).eval()
    hsd = depth_anything_hf_state_dict(sd, depth)
    missing, unexpected = model.load_state_dict(hsd, strict=False)
    missing = [k for k in missing if "mask_token" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    if upstream_pos_embed:
        emb = model.backbone.embeddings
        emb.interpolate_pos_encoding = types.MethodType(_upstream_pos_embed, emb)
    return model


def depth_anything_hf_forward(sd, x, taps=None, max_depth=0.0, upstream_pos_embed=True):
    """x [B,3,h,w] ImageNet-normalised, multiples of 14 -> [B,h,w] like ``oracle.depth_anything_v2.model_forward``."""
    model = depth_anything_hf(sd, taps, max_depth, upstream_pos_embed)
    with torch.no_grad():
        return model(pixel_values=x).predicted_depth
