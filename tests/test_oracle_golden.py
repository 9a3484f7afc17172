"""The oracle against the committed fixtures (outputs of the reference itself, tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, sd_checksum
from oracle import seam_blending as OS
from oracle import swin_unet as O

TOL = 2e-4   # fp32 op-order noise of two CPU evaluations of a 14-block net (measured ~3e-5)


def test_seam_configs_match_reference():
    cases = json.load(open(os.path.join(GOLDEN, "seam_configs.json")))
    assert len(cases) >= 10
    for case in cases:
        h, w, s, o, t, b = case["args"]
        cfg = OS.create_config(h, w, s, o, t, b)
        ref = dict(case["config"])
        ref["pad"] = tuple(ref["pad"])
        assert cfg == ref, case["args"]
        if b > 0:
            f = OS.blend_filter(s, o, t, b, 1)[0]
            mid = f.shape[0] // 2
            assert [int(v) for v in f[mid, :b + 1].view(torch.int32)] == case["ramp_bits"]   # bit-exact ramp
            assert float(f.double().sum()) == case["filter_sum"]
            assert [float(v) for v in f[:b + 1, :b + 1].reshape(-1)] == case["corner"]


def test_blend_filter_is_a_pyramid_not_a_product():
    f = OS.blend_filter(2, 16, 64, 8, 3)
    assert f.shape == (3, 96, 96)
    assert torch.equal(f[0], f[0].t()) and torch.equal(f[0], f[0].flip(0)) and torch.equal(f[0], f[2])
    assert f[0, 0, 0].item() == pytest.approx(1 / 9) and f[0, 40, 40].item() == 1.0
    assert f[0, 2, 50].item() == pytest.approx(3 / 9) and f[0, 2, 1].item() == pytest.approx(2 / 9)


@pytest.mark.parametrize("tag,sf", [("1x", 1), ("2x", 2), ("4x", 4)])
def test_swin_forward_matches_reference(golden_swin, tag, sf):
    sd = O.random_state_dict(100 + sf, sf)
    assert sd_checksum(sd) == pytest.approx(float(golden_swin["sdsum_" + tag]), rel=1e-12), "RNG stream differs"
    x = torch.from_numpy(golden_swin["x"])
    name = {1: "waifu2x.swin_unet_1x", 2: "waifu2x.swin_unet_2x", 4: "waifu2x.swin_unet_4x"}[sf]
    y = O.model_forward(sd, x, name)
    ref = torch.from_numpy(golden_swin["y_" + tag])
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() < TOL
    assert 0.05 < ref.std().item() < 0.45, "fixture is saturated; PSNR would be meaningless"


@pytest.mark.parametrize("tag,sf", [("1x", 1), ("2x", 2), ("4x", 4)])
def test_swin_forward_matches_hf_backed_reference(golden_swin_hf, tag, sf):
    """``swin_unet_hf.npz`` = the reference's U-Net over HuggingFace's ``SwinLayer`` instead of ``oracle/tv_swin_block.py``
    (tests/golden/make_golden_hf.py): the oracle's whole-net restatement agrees with an output that our block restatement
    did not produce."""
    sd = O.random_state_dict(100 + sf, sf)
    assert sd_checksum(sd) == pytest.approx(float(golden_swin_hf["sdsum_" + tag]), rel=1e-12)
    name = {1: "waifu2x.swin_unet_1x", 2: "waifu2x.swin_unet_2x", 4: "waifu2x.swin_unet_4x"}[sf]
    y = O.model_forward(sd, torch.from_numpy(golden_swin_hf["x"]), name)
    assert (y - torch.from_numpy(golden_swin_hf["y_" + tag])).abs().max().item() < TOL


def test_swin_downscaled_matches_reference(golden_swin):
    sd = O.random_state_dict(104, 4)
    x = torch.from_numpy(golden_swin["x"])
    for key, f in (("y_4x_to2x", 2), ("y_4x_to1x", 4)):
        y = O.model_forward(sd, x, downscale_factor=f)
        assert (y - torch.from_numpy(golden_swin[key])).abs().max().item() < TOL


def test_swin_forward_112_and_tiled_render(golden_swin):
    sd = O.random_state_dict(102, 2)
    y = O.model_forward(sd, torch.from_numpy(golden_swin["x_112"]))
    assert (y - torch.from_numpy(golden_swin["y_2x_112"])).abs().max().item() < TOL
    img = torch.from_numpy(golden_swin["img"])
    ref = torch.from_numpy(golden_swin["render_2x_t64_b4"])
    fn = lambda mb: O.model_forward(sd, mb)   # noqa: E731
    out = OS.tiled_render(img, fn, 2, 16, 8, 64, 4)
    assert out.shape == ref.shape == (3, 200, 260)
    assert (out - ref).abs().max().item() < TOL
    # closed form (what the HIP stitcher evaluates) == cumulative form
    out2 = OS.tiled_render_closed_form(img, fn, 2, 16, 8, 64, 4)
    assert (out2 - out).abs().max().item() < 2e-6


def test_tile_size_validator():
    assert [t for t in range(1, 300) if O.valid_tile_size(t)] == [64, 112, 160, 208, 256]
    assert O.find_valid_tile_size(256) == 256 and O.find_valid_tile_size(255) == 208
    assert O.find_valid_tile_size(640) == 640
    with pytest.raises(ValueError):
        O.find_valid_tile_size(63)


def test_stitch_identity_model_reproduces_nearest_upscale():
    """Known-answer test for the stitcher alone: a 'model' that returns the centre crop upscaled by nearest."""
    scale, offset, blend, tile = 2, 16, 8, 64
    x = torch.rand(3, 75, 141)

    def model(mb):
        up = torch.nn.functional.interpolate(mb, scale_factor=scale, mode="nearest")
        return up[:, :, offset:-offset, offset:-offset]

    out = OS.tiled_render(x, model, scale, offset, blend, tile, 4)
    expect = torch.nn.functional.interpolate(x[None], scale_factor=scale, mode="nearest")[0]
    assert (out - expect).abs().max().item() < 1e-6


def test_swin_8x_matches_reference():
    """waifu2x.swin_unet_8x: the two-layer ToImage head + pixel_shuffle(8) (fixture stored as fp16)."""
    import os
    import numpy as np
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "swin_unet_8x.npz"))
    sd = O.random_state_dict(108, 8)
    assert sd_checksum(sd) == pytest.approx(float(g["sdsum"]), rel=1e-12)
    y = O.model_forward(sd, torch.from_numpy(g["x"]), "waifu2x.swin_unet_8x")
    ref = torch.from_numpy(g["y"]).float()
    assert y.shape == ref.shape == (1, 3, 384, 384) and (y - ref).abs().max().item() < 1e-3


def test_swin_4xl_matches_reference():
    """waifu2x.swin_unet_4xl (base_dim 192, 12 heads, LayerNormNoBias; reference swin_unet.py:390-394) and the LayerNorm variant of
    the 2x net: the oracle's norm1 / norm2 handling against the reference's own outputs (fixtures stored as fp16)."""
    import os
    import numpy as np
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "swin_unet_4xl.npz"))
    x = torch.from_numpy(g["x"])
    sd = O.random_state_dict(204, 4, base_dim=192, layer_norm=True)
    assert sd_checksum(sd) == pytest.approx(float(g["sdsum"]), rel=1e-12)
    y = O.model_forward(sd, x, "waifu2x.swin_unet_4x")
    ref = torch.from_numpy(g["y"]).float()
    assert y.shape == ref.shape == (1, 3, 192, 192) and (y - ref).abs().max().item() < 1e-3
    sd2 = O.random_state_dict(205, 2, base_dim=96, layer_norm=True)
    assert sd_checksum(sd2) == pytest.approx(float(g["sdsum2"]), rel=1e-12)
    y2 = O.model_forward(sd2, x, "waifu2x.swin_unet_2x")
    ref2 = torch.from_numpy(g["y2_ln"]).float()
    assert y2.shape == ref2.shape == (1, 3, 96, 96) and (y2 - ref2).abs().max().item() < 1e-3
