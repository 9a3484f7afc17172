"""waifu2x ``swin_unet_v2`` (1x / 2x / 4x): the oracle restatement against the reference's own modules (fixture from
``make_golden.py::gen_swin_v2``), and the product models' contract (registry names, geometry, state-dict layout) against the
live reference.  The HIP engine itself is held to the oracle in ``test_gpu_swin_v2.py``."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, sd_checksum
from oracle import swin_unet_v2 as OV


def _reference_init_state_dict(tag, seed):
    """The fixture's weights: the reference constructor under ``torch.manual_seed`` + ``OV.randomize``.  Rebuilt here only when
    the reference tree is present; elsewhere the test is skipped (the oracle needs the exact weights)."""
    from oracle import refstub
    if not refstub.reference_available():
        pytest.skip("needs /root/reference to rebuild the seeded constructor weights")
    refstub.install()
    from waifu2x.models import swin_unet_v2 as RV
    cls = {"1x": RV.SwinUNet1xV2, "2x": RV.SwinUNet2xV2, "4x": RV.SwinUNet4xV2}[tag]
    torch.manual_seed(seed)
    return OV.randomize(cls().eval().state_dict(), seed + 100)


@pytest.mark.parametrize("tag,seed,scale,offset", [("1x", 11, 1, 9), ("2x", 12, 2, 18), ("4x", 13, 4, 36)])
def test_oracle_matches_reference(tag, seed, scale, offset):
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "swin_unet_v2.npz")).items()}
    sd = _reference_init_state_dict(tag, seed)
    assert sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point}) == pytest.approx(float(g["sdsum_" + tag]), rel=1e-12)
    x = g["x"]
    raw = OV.model_forward(sd, x, scale, raw=True)
    assert raw.shape == g["raw_" + tag].shape == (2, 3, 64 * scale - 2 * offset, 64 * scale - 2 * offset)
    assert (raw - g["raw_" + tag]).abs().max().item() < 2e-4
    y = OV.model_forward(sd, x, scale)
    assert (y - g["y_" + tag]).abs().max().item() < 2e-4
    sat = ((g["y_" + tag] <= 0) | (g["y_" + tag] >= 1)).float().mean().item()
    assert sat < 0.35 and g["y_" + tag].std().item() > 0.05            # a meaningful fixture: not clamped away


NAMES = {"1x": ("waifu2x.swin_unet_v2_1x", ("waifu2x.winc_unet_1x", "waifu2x.swin_unet_1x_v2")),
         "2x": ("waifu2x.swin_unet_v2_2x", ("waifu2x.winc_unet_2x",)),
         "4x": ("waifu2x.swin_unet_v2_4x", ("waifu2x.winc_unet_4x",))}


@pytest.mark.parametrize("tag,scale", [("1x", 1), ("2x", 2), ("4x", 4)])
def test_product_models_registered_with_reference_geometry(tag, scale):
    from nunif_amd.nunif.models import create_model
    import nunif_amd.waifu2x.utils  # noqa: F401
    name, aliases = NAMES[tag]
    m = create_model(name)
    assert (m.i2i_scale, m.i2i_offset, m.i2i_blend_size) == (scale, 9 * scale, 4 * scale)
    for a in aliases:
        assert type(create_model(a)) is type(m)
    assert m.find_valid_tile_size(256) == 256 and m.find_valid_tile_size(128) == 112 and m.find_valid_tile_size(64) == 64
    # a fresh model is the nearest-neighbour upscaler (scale_bias 0) and has no CPU path
    assert float(m.state_dict()["unet.to_image.scale_bias"]) == 0.0
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 3, 64, 64))
    from nunif_amd import synthetic
    sd = synthetic.swin_unet_v2_state_dict(5, scale)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    bad = dict(sd)
    bad.pop("unet.patch.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    if tag == "4x":
        d2, d1 = m.to_2x(), m.to_1x(shared=False)
        assert (d2.i2i_scale, d2.i2i_offset, d2.i2i_blend_size) == (2, 18, 8)
        assert (d1.i2i_scale, d1.i2i_offset, d1.i2i_blend_size) == (1, 9, 16)
        assert d2.net4x is m and d1.net4x is not m


@pytest.mark.parametrize("tag", ["1x", "2x", "4x"])
def test_state_dict_layout_equals_reference_constructor(tag):
    """Keys, shapes and dtypes (including the score-bias ``index`` / ``delta`` buffers and their VALUES) of the product model
    and of the synthetic generator against the reference constructor; geometry attributes and the validator likewise."""
    from oracle import refstub
    if not refstub.reference_available():
        pytest.skip("needs /root/reference")
    refstub.install()
    from waifu2x.models import swin_unet_v2 as RV
    from nunif_amd.nunif.models import create_model
    import nunif_amd.waifu2x.utils  # noqa: F401
    ref = {"1x": RV.SwinUNet1xV2, "2x": RV.SwinUNet2xV2, "4x": RV.SwinUNet4xV2}[tag]().eval()
    m = create_model(NAMES[tag][0])
    rsd, sd = ref.state_dict(), m.state_dict()
    assert list(rsd.keys()) == list(sd.keys())
    for k in rsd:
        assert rsd[k].shape == sd[k].shape and rsd[k].dtype == sd[k].dtype, k
        if k.endswith("relative_bias.index") or k.endswith("relative_bias.delta") or k.endswith("resampling.weight"):
            assert torch.equal(rsd[k], sd[k]), k
    assert (ref.i2i_scale, ref.i2i_offset, ref.i2i_blend_size, ref.i2i_in_channels) == \
        (m.i2i_scale, m.i2i_offset, m.i2i_blend_size, m.i2i_in_channels)
    assert ref.name == m.name and tuple(ref.name_alias) == tuple(m.name_alias)
    for size in (64, 100, 112, 128, 256, 400, 640):
        assert ref.find_valid_tile_size(size) == m.find_valid_tile_size(size)
    with pytest.raises(ValueError):
        ref.find_valid_tile_size(32)
    with pytest.raises(ValueError):
        m.find_valid_tile_size(32)
    ref.load_state_dict(sd, strict=True)          # a product state dict loads into the reference, and back
    m.load_state_dict(rsd, strict=True)
    if tag == "4x":
        r2, p2 = ref.to_2x(), m.to_2x()
        assert (r2.i2i_scale, r2.i2i_offset, r2.i2i_blend_size) == (p2.i2i_scale, p2.i2i_offset, p2.i2i_blend_size)
        r1, p1 = ref.to_1x(), m.to_1x()
        assert (r1.i2i_scale, r1.i2i_offset, r1.i2i_blend_size) == (p1.i2i_scale, p1.i2i_offset, p1.i2i_blend_size)
