"""Pins ``oracle/tv_swin_block.py`` (the restatement of torchvision's ``SwinTransformerBlock`` V1 that ~90 % of the swin_unet
FLOPs run through) against the REAL class wherever torchvision is importable.

torchvision is an external dependency of the reference (``requirements-torch.txt``: torchvision==0.22.1, imported at
``waifu2x/models/swin_unet.py:9-12``); it is not installed in the build container or on the GPU box, so this file SKIPS
there — DESIGN.md §2 says which box ran it.  Same state dict in both modules, fp32 on CPU, the cases the waifu2x nets hit:
no shift, shift 3, a map that needs padding to the window, 6 and 12 heads, the window covering the whole map.
"""
import pytest
import torch

tv = pytest.importorskip("torchvision", reason="torchvision (external dependency of the reference) is not installed here")
from torchvision.models.swin_transformer import SwinTransformerBlock as TVBlock  # noqa: E402

from oracle.tv_swin_block import SwinTransformerBlock as OracleBlock  # noqa: E402

CASES = [
    # dim, heads, H, W, shift
    (96, 6, 12, 12, 0),
    (96, 6, 12, 12, 3),
    (192, 6, 18, 24, 3),
    (192, 12, 12, 12, 3),       # swin_unet_4xl
    (96, 6, 14, 10, 3),         # needs padding to a multiple of the 6 x 6 window
    (96, 6, 6, 6, 3),           # window covers the map: torchvision disables the shift
]


@pytest.mark.parametrize("dim,heads,H,W,shift", CASES)
def test_oracle_block_equals_torchvision(dim, heads, H, W, shift):
    torch.manual_seed(dim + heads + H + shift)
    no_norm = lambda d: torch.nn.Identity()      # noqa: E731   (waifu2x/models/swin_unet.py:16-17 NO_NORM_LAYER)
    ref = TVBlock(dim, heads, window_size=[6, 6], shift_size=[shift, shift], mlp_ratio=2.0, dropout=0.0,
                  attention_dropout=0.0, stochastic_depth_prob=0.0, norm_layer=no_norm).eval()
    with torch.no_grad():
        ref.attn.relative_position_bias_table.normal_(0, 0.5)
        for p in ref.parameters():
            if p.ndim == 1:
                p.normal_(0, 0.1)
    mine = OracleBlock(dim, heads, window_size=[6, 6], shift_size=[shift, shift], mlp_ratio=2.0, norm_layer=no_norm).eval()
    missing, unexpected = mine.load_state_dict(ref.state_dict(), strict=False)
    assert not [k for k in missing if "relative_position_index" not in k], missing
    assert not [k for k in unexpected if "relative_position_index" not in k], unexpected
    x = torch.randn(2, H, W, dim)
    with torch.no_grad():
        a, b = ref(x), mine(x)
    assert a.shape == b.shape
    assert torch.allclose(a, b, rtol=0, atol=2e-5), (a - b).abs().max().item()


def test_state_dict_keys_match_torchvision():
    ref = TVBlock(96, 6, window_size=[6, 6], shift_size=[3, 3], mlp_ratio=2.0)
    mine = OracleBlock(96, 6, window_size=[6, 6], shift_size=[3, 3], mlp_ratio=2.0)
    strip = lambda sd: sorted(k for k in sd if "relative_position_index" not in k)      # noqa: E731
    assert strip(ref.state_dict()) == strip(mine.state_dict())
    for k in strip(ref.state_dict()):
        assert ref.state_dict()[k].shape == mine.state_dict()[k].shape, k
