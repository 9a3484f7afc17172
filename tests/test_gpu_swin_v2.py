"""swin_unet_v2 (winc_unet) HIP engine against the oracle (``oracle/swin_unet_v2.py``, itself pinned to the reference's own
modules by ``tests/golden/swin_unet_v2.npz`` / ``test_swin_unet_v2_oracle.py``).

The family has no released checkpoint, so the weights are the seeded synthetic ones of ``nunif_amd.synthetic`` (key layout
checked against the reference constructor in the CPU suite).  Tolerance: float PSNR >= 50 dB on the clamped [0,1] output
(BASELINE.json north_star) and a bound on the un-clamped residual image; fp16 maps with fp32 accumulation vs fp32 end to end.
"""
import pytest
import torch

from conftest import psnr, synth_image
from oracle import swin_unet_v2 as OV

pytestmark = pytest.mark.gpu
NAMES = {1: "waifu2x.swin_unet_v2_1x", 2: "waifu2x.swin_unet_v2_2x", 4: "waifu2x.swin_unet_v2_4x"}


def make_model(sf, seed):
    from nunif_amd import synthetic
    from nunif_amd.nunif.models import create_model
    import nunif_amd.waifu2x.utils  # noqa: F401  (registers the models)
    sd = synthetic.swin_unet_v2_state_dict(seed, sf)
    m = create_model(NAMES[sf]).eval()
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0"), sd


@pytest.mark.parametrize("sf", [1, 2, 4])
def test_forward_matches_oracle(hiplib, sf, capsys):
    """One batch of two 64 x 64 tiles (level maps 48 / 24; the 6 x 6 and every shifted 8 x 8 window layout is active, and the
    24 x 24 level has border windows made mostly of zero padding)."""
    m, sd = make_model(sf, 40 + sf)
    x = torch.stack([synth_image(301, 3, 64, 64), synth_image(302, 3, 64, 64)])
    raw_ref = OV.model_forward(sd, x, sf, raw=True)
    y = m(x.to("cuda:0")).cpu()
    raw = m.engine().forward(x.to("cuda:0"), clamp=False).cpu()
    assert y.shape == raw_ref.shape == (2, 3, 64 * sf - 18 * sf, 64 * sf - 18 * sf)
    err = (raw - raw_ref).abs()
    with capsys.disabled():
        print(f"\nswin_unet_v2 {sf}x: PSNR {psnr(y, raw_ref.clamp(0, 1)):.2f} dB, raw max err {err.max().item():.2e}, "
              f"rms {err.pow(2).mean().sqrt().item():.2e} (raw std {raw_ref.std().item():.3f})")
    assert psnr(y, raw_ref.clamp(0, 1)) >= 50.0
    assert err.max().item() < 2e-2 and err.pow(2).mean().sqrt().item() < 3e-3
    assert torch.equal(y, raw.clamp(0, 1))


def test_larger_tile_and_batch_of_one(hiplib):
    """Tile 112 (the reference's training tile; maps 96 / 48, IR at 56) with a single tile."""
    m, sd = make_model(2, 52)
    x = synth_image(303, 3, 112, 112)[None]
    y_ref = OV.model_forward(sd, x, 2)
    y = m(x.to("cuda:0")).cpu()
    assert y.shape == (1, 3, 188, 188)
    assert psnr(y, y_ref) >= 50.0


def test_tiled_render_and_downscaled(hiplib):
    """Through the reference's own entry point: tiled_render over a frame that needs several tiles (generic gather -> model ->
    stitch route, offset 18 / blend 8), against the same loop run with the oracle as the model; and the 4x -> 2x wrapper."""
    from nunif_amd.nunif.utils.seam_blending import SeamBlending
    from oracle import seam_blending as OS
    m, sd = make_model(2, 61)
    x = synth_image(304, 3, 150, 200)
    y = SeamBlending.tiled_render(x.to("cuda:0"), m, tile_size=112, batch_size=2).cpu()
    y_ref = OS.tiled_render(x, lambda t: OV.model_forward(sd, t, 2), scale=2, offset=18, tile_size=112, blend_size=8, batch_size=2)
    assert y.shape == y_ref.shape == (3, 300, 400)
    assert psnr(y, y_ref) >= 50.0
    m4, sd4 = make_model(4, 62)
    d = m4.to_2x()
    assert (d.i2i_scale, d.i2i_offset, d.i2i_blend_size) == (2, 18, 8)
    t = torch.stack([synth_image(305, 3, 64, 64)])
    z = d(t.to("cuda:0")).cpu()
    z4 = OV.model_forward(sd4, t, 4)
    z_ref = torch.nn.functional.interpolate(z4, size=(z4.shape[-2] // 2, z4.shape[-1] // 2), mode="bicubic", align_corners=False,
                                            antialias=True).clamp(0, 1)
    assert z.shape == z_ref.shape and psnr(z, z_ref) >= 50.0


def test_invalid_tile_size_is_rejected(hiplib):
    m, _ = make_model(1, 71)
    with pytest.raises(Exception):
        m(torch.zeros(1, 3, 72, 72, device="cuda:0"))
