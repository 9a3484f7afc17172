"""waifu2x ``swin_unet_v2`` (1x / 2x / 4x): the oracle restatement against the reference's own modules (fixture from
``make_golden.py::gen_swin_v2``).  The HIP engine does not carry this family yet; the registered names raise."""
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


def test_product_names_are_not_silently_something_else():
    from nunif_amd.nunif.models import create_model
    import nunif_amd.waifu2x.models.swin_unet  # noqa: F401
    for name in ("waifu2x.swin_unet_v2_2x", "waifu2x.winc_unet_2x"):
        with pytest.raises(Exception):
            create_model(name)
