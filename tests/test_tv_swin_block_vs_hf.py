"""Pins ``oracle/tv_swin_block.py`` against an INDEPENDENT implementation of the Swin V1 block: HuggingFace
``transformers.models.swin.modeling_swin.SwinLayer`` (installed in the build container).

torchvision (the class the reference imports at ``waifu2x/models/swin_unet.py:9-12,26-36``) is installed nowhere
this code runs, so ``tests/test_tv_swin_block_live.py`` skips; the HuggingFace layer implements the same published
algorithm (Liu et al. 2021: window partition, cyclic shift, 9-region mask of -100, relative position bias table +
index, softmax(QK^T/sqrt(d) + bias + mask) V, proj; then the MLP) with a different code base and parameter layout:

    torchvision key                          HuggingFace key
    attn.qkv.{weight,bias}   [3C, C]      -> attention.{q,k,v}_proj.{weight,bias}   (rows 0:C, C:2C, 2C:3C)
    attn.proj                              -> attention.o_proj
    attn.relative_position_bias_table      -> attention.relative_position_bias.relative_position_bias_table
    mlp.0 / mlp.3                          -> mlp.fc1 / mlp.fc2
    norm1 / norm2                          -> layernorm_before / layernorm_after (nn.Identity for NO_NORM_LAYER,
                                              bias-free LayerNorm for LayerNormNoBias: nunif/modules/norm.py:17-21)

Where the two libraries DIFFER is outside what the waifu2x nets reach (asserted by ``swin_unet.py:183``: maps are
multiples of the window): HF turns the shift off when min(H, W) <= window (torchvision does it per axis) — the cases
below keep both axes above the window or both equal to it.  A deliberately wrong shift is checked to FAIL, so the
comparison is not vacuous.
"""
import pytest
import torch
import torch.nn as nn

hf = pytest.importorskip("transformers.models.swin.modeling_swin",
                         reason="HuggingFace transformers is not installed here")

from oracle.tv_swin_block import SwinTransformerBlock as OracleBlock  # noqa: E402
from oracle.tv_swin_block import relative_position_index  # noqa: E402


def _no_norm(dim):
    return nn.Identity()


def _ln_nobias(dim):
    return nn.LayerNorm(dim, eps=1e-5, bias=False)


def _hf_layer(oracle, dim, heads, shift, H, W, norm):
    cfg = hf.SwinConfig(window_size=6, mlp_ratio=2.0, qkv_bias=True, hidden_act="gelu", hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5)
    cfg._attn_implementation = "eager"
    layer = hf.SwinLayer(cfg, dim, (H, W), heads, drop_path_rate=0.0, shift_size=shift).eval()
    sd = oracle.state_dict()
    C = dim
    with torch.no_grad():
        att = layer.attention
        for i, p in enumerate((att.q_proj, att.k_proj, att.v_proj)):
            p.weight.copy_(sd["attn.qkv.weight"][i * C:(i + 1) * C])
            p.bias.copy_(sd["attn.qkv.bias"][i * C:(i + 1) * C])
        att.o_proj.weight.copy_(sd["attn.proj.weight"])
        att.o_proj.bias.copy_(sd["attn.proj.bias"])
        att.relative_position_bias.relative_position_bias_table.copy_(sd["attn.relative_position_bias_table"])
        layer.mlp.fc1.weight.copy_(sd["mlp.0.weight"])
        layer.mlp.fc1.bias.copy_(sd["mlp.0.bias"])
        layer.mlp.fc2.weight.copy_(sd["mlp.3.weight"])
        layer.mlp.fc2.bias.copy_(sd["mlp.3.bias"])
        if norm == "none":
            layer.layernorm_before = nn.Identity()
            layer.layernorm_after = nn.Identity()
        else:
            for mine, name in ((layer.layernorm_before, "norm1"), (layer.layernorm_after, "norm2")):
                mine.weight.copy_(sd[name + ".weight"])
                mine.bias.zero_()
    return layer


def _make_oracle(dim, heads, shift, norm, seed):
    torch.manual_seed(seed)
    blk = OracleBlock(dim, heads, window_size=[6, 6], shift_size=[shift, shift], mlp_ratio=2.0,
                      norm_layer=_no_norm if norm == "none" else _ln_nobias).eval()
    with torch.no_grad():
        blk.attn.relative_position_bias_table.normal_(0, 0.5)
        for p in blk.parameters():
            if p.ndim == 1:
                p.normal_(0, 0.1)
        if norm != "none":
            blk.norm1.weight.normal_(1.0, 0.2)
            blk.norm2.weight.normal_(1.0, 0.2)
    return blk


CASES = [
    # dim, heads, H, W, shift, norm
    (96, 6, 12, 12, 0, "none"),        # level 1, W-MSA
    (96, 6, 12, 12, 3, "none"),        # level 1, SW-MSA
    (96, 6, 6, 6, 3, "none"),          # window covers the map: shift disabled in both libraries
    (96, 6, 14, 16, 3, "none"),        # needs padding to a multiple of the window (never hit by waifu2x, tested anyway)
    (192, 6, 24, 18, 3, "none"),       # levels 2 / 3 (head_dim 32), non-square
    (192, 6, 18, 24, 0, "none"),
    (192, 12, 12, 12, 3, "ln"),        # swin_unet_4xl: 12 heads + LayerNormNoBias
    (96, 6, 18, 12, 3, "ln"),          # SwinUNet2x(layer_norm=True)
]


@pytest.mark.parametrize("dim,heads,H,W,shift,norm", CASES)
def test_oracle_block_equals_huggingface_swin_layer(dim, heads, H, W, shift, norm):
    blk = _make_oracle(dim, heads, shift, norm, seed=dim + heads + H + 7 * shift)
    layer = _hf_layer(blk, dim, heads, shift, H, W, norm)
    x = torch.randn(2, H, W, dim)
    with torch.no_grad():
        a = blk(x)
        b = layer(x.reshape(2, H * W, dim), (H, W))[0].reshape(2, H, W, dim)
    err = (a - b).abs().max().item()
    assert err <= 2e-5, err


def test_relative_position_index_equals_huggingface():
    rpb = hf.SwinRelativePositionBias(6, (6, 6))
    assert torch.equal(rpb.relative_position_index.view(-1), relative_position_index(6, 6))
    rpb = hf.SwinRelativePositionBias(2, (4, 8))
    assert torch.equal(rpb.relative_position_index.view(-1), relative_position_index(4, 8))


def test_comparison_is_not_vacuous():
    """A wrong shift (HF layer built with shift 2 against the oracle's 3) must NOT match."""
    blk = _make_oracle(96, 6, 3, "none", seed=5)
    layer = _hf_layer(blk, 96, 6, 2, 12, 12, "none")
    x = torch.randn(1, 12, 12, 96)
    with torch.no_grad():
        a = blk(x)
        b = layer(x.reshape(1, 144, 96), (12, 12))[0].reshape(1, 12, 12, 96)
    assert (a - b).abs().max().item() > 1e-2
