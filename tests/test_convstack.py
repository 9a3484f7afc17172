"""waifu2x.vgg_7 / waifu2x.upconv_7: oracle vs the reference fixture (CPU), HIP engine vs fixture (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr, sd_checksum
from oracle import cunet as OC

CASES = (("vgg_7", 601, 1, 7), ("upconv_7", 602, 2, 14))


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "convstack.npz")).items()}


@pytest.mark.parametrize("tag,seed,scale,offset", CASES)
def test_oracle_matches_reference_fixture(g, tag, seed, scale, offset):
    sd = OC.conv_stack_state_dict(seed, tag)
    assert sd_checksum(sd) == pytest.approx(float(g[tag + "_sdsum"]), rel=1e-12)
    z = OC.conv_stack_forward(sd, g["x"])
    assert z.shape == g[tag + "_z"].shape == (2, 3, 64 * scale - 2 * offset, 64 * scale - 2 * offset)
    assert (z - g[tag + "_z"]).abs().max().item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag,seed,scale,offset", CASES)
def test_hip_conv_stack(hiplib, g, tag, seed, scale, offset):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.nunif.utils.render import tiled_render
    from nunif_amd.waifu2x.models import vgg_7  # noqa: F401
    m = create_model("waifu2x." + tag).eval()
    assert (m.i2i_scale, m.i2i_offset) == (scale, offset)
    m.load_state_dict(OC.conv_stack_state_dict(seed, tag), strict=True)
    m = m.to("cuda:0")
    with torch.no_grad():
        z = m(g["x"].to("cuda:0"))
        assert z.shape == g[tag + "_z"].shape
        p = psnr(z.cpu(), g[tag + "_z"])
        assert p >= 50.0, (tag, p)
        y = tiled_render(g["frame"].to("cuda:0"), m, tile_size=64, batch_size=4)
        assert y.shape == g[tag + "_render"].shape
        p = psnr(y.cpu(), g[tag + "_render"])
        assert p >= 50.0, (tag, "render", p)
    with pytest.raises(RuntimeError):
        m.load_state_dict({"net.0.weight": torch.zeros(1)}, strict=True)
    with pytest.raises(RuntimeError):
        create_model("waifu2x." + tag).eval()(g["x"])            # CPU-resident model: no fallback
