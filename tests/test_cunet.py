"""CUNet: oracle vs the reference fixture (CPU) and HIP engine vs oracle / fixture (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr, sd_checksum, synth_image
from oracle import cunet as OC
from oracle import seam_blending as OS


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "cunet.npz")).items()}


def test_oracle_matches_reference_fixture(g):
    sd = OC.random_state_dict(201)
    assert sd_checksum(sd) == pytest.approx(float(g["sdsum"]), rel=1e-12)
    y = OC.model_forward(sd, g["x"])
    assert y.shape == g["y"].shape == (2, 3, 40, 40) and (y - g["y"]).abs().max().item() < 1e-5
    assert (OC.model_forward(sd, g["x"][:1], no_clip=True) - g["y_no_clip"]).abs().max().item() < 1e-5
    out = OS.tiled_render(g["img"], lambda mb: OC.model_forward(sd, mb), 1, 28, 0, 96, 4)
    assert out.shape == g["render_t96_b4"].shape and (out - g["render_t96_b4"]).abs().max().item() < 1e-5
    assert 0.05 < g["y"].std().item() < 0.4 and float((g["y"] <= 0).float().mean()) < 0.15


def test_oracle_upcunet_matches_reference_fixture(g):
    sd = OC.random_state_dict(203, up=True)
    assert sd_checksum(sd) == pytest.approx(float(g["up_sdsum"]), rel=1e-12)
    y = OC.model_forward(sd, g["x"], no_clip=True)
    assert y.shape == g["up_y"].shape == (2, 3, 120, 120) and (y - g["up_y"]).abs().max().item() < 1e-5
    assert (OC.model_forward(sd, g["x"][:1]) - g["up_y_clip"]).abs().max().item() < 1e-5
    out = OS.tiled_render(g["up_img"], lambda mb: OC.model_forward(sd, mb, no_clip=True), 2, 36, 0, 64, 5)
    assert out.shape == g["up_render_t64_b5"].shape == (3, 200, 260)
    assert (out - g["up_render_t64_b5"]).abs().max().item() < 1e-5
    assert 0.05 < g["up_y"].std().item() < 0.4 and float((g["up_y"] <= 0).float().mean()) < 0.15


def test_geometry():
    assert OC.GEOMETRY["waifu2x.upcunet"] == (2, 36, 0)
    assert OC.GEOMETRY["waifu2x.cunet"] == (1, 28, 0)
    assert [t for t in range(60, 80) if OC.valid_tile_size(t)] == [60, 64, 68, 72, 76]
    cfg = OS.create_config(512, 512, 1, 28, 256, 0)
    assert (cfg["h_blocks"], cfg["w_blocks"], cfg["input_tile_step"], cfg["pad"]) == (3, 3, 200, (28, 116, 28, 116))


@pytest.mark.gpu
def test_hip_forward_and_render(hiplib, g):
    from nunif_amd.waifu2x.models.cunet import CUNet
    from nunif_amd.nunif.utils.render import tiled_render
    sd = OC.random_state_dict(201)
    m = CUNet().eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    assert (m.i2i_scale, m.i2i_offset, m.i2i_blend_size) == (1, 28, None)
    y = m(g["x"].to("cuda:0")).cpu()
    assert y.shape == g["y"].shape and float(y.min()) >= 0 and float(y.max()) <= 1
    assert psnr(y, g["y"]) >= 50.0, psnr(y, g["y"])            # vs the reference's own output
    y1 = m(g["x"][1:2].to("cuda:0")).cpu()
    assert torch.equal(y1, y[1:2]), "result depends on the minibatch (SE pooling must stay per tile)"
    out = tiled_render(g["img"], m, tile_size=96, batch_size=4)
    assert psnr(out.cpu(), g["render_t96_b4"]) >= 50.0
    assert torch.equal(out, tiled_render(g["img"], m, tile_size=96, batch_size=9))
    m2 = CUNet(no_clip=True).eval()
    m2.load_state_dict(sd)
    assert psnr(m2.to("cuda:0")(g["x"][:1].to("cuda:0")).cpu(), g["y_no_clip"]) >= 50.0


@pytest.mark.gpu
def test_hip_upcunet_forward_and_render(hiplib, g):
    from nunif_amd.waifu2x.models.cunet import UpCUNet
    from nunif_amd.nunif.utils.render import tiled_render
    sd = OC.random_state_dict(203, up=True)
    m = UpCUNet(no_clip=True).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    assert (m.i2i_scale, m.i2i_offset, m.i2i_blend_size) == (2, 36, None)
    y = m(g["x"].to("cuda:0")).cpu()
    assert y.shape == g["up_y"].shape and float(y.min()) >= 0 and float(y.max()) <= 1
    assert psnr(y, g["up_y"]) >= 50.0, psnr(y, g["up_y"])      # vs the reference's own output
    mc = UpCUNet().eval()
    mc.load_state_dict(sd)
    assert psnr(mc.to("cuda:0")(g["x"][:1].to("cuda:0")).cpu(), g["up_y_clip"]) >= 50.0
    out = tiled_render(g["up_img"], m, tile_size=64, batch_size=5)
    assert out.shape == (3, 200, 260) and psnr(out.cpu(), g["up_render_t64_b5"]) >= 50.0
    assert torch.equal(out, tiled_render(g["up_img"], m, tile_size=64, batch_size=12))
    # a 256 tile (the default): 2*256 - 72 = 440, interior vs the oracle
    x = synth_image(62, 3, 256, 256)[None]
    z = m(x.to("cuda:0")).cpu()
    assert z.shape == (1, 3, 440, 440) and psnr(z, OC.model_forward(sd, x, no_clip=True)) >= 50.0


@pytest.mark.gpu
def test_hip_config1_512_tile256(hiplib):
    """BASELINE config 1 geometry: cunet on a 512x512 image, tile 256 (9 tiles); oracle on the full image is ~10 s
    of CPU, so compare one tile's interior and check determinism / range / shape for the frame."""
    from nunif_amd.waifu2x.models.cunet import CUNet
    from nunif_amd.nunif.utils.render import tiled_render
    sd = OC.random_state_dict(202)
    m = CUNet().eval()
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    img = synth_image(61, 3, 512, 512)
    out = tiled_render(img, m, tile_size=256, batch_size=4)
    assert out.shape == (3, 512, 512) and torch.equal(out, tiled_render(img, m, tile_size=256, batch_size=9))
    cfg = OS.create_config(512, 512, 1, 28, 256, 0)
    xp = torch.nn.functional.pad(img[None], cfg["pad"], mode="replicate")[0]
    z = OC.model_forward(sd, xp[:, 200:456, 200:456][None])[0]        # tile (1,1) -> output [200:400)
    assert psnr(out[:, 200:400, 200:400].cpu(), z) >= 50.0
    with pytest.raises(Exception):
        m(torch.rand(1, 3, 62, 62).to("cuda:0"))                     # not a multiple of 4


@pytest.mark.gpu
def test_hip_cunet_1080p_whole_frame_minibatch_equals_small_minibatches(hiplib):
    """bench.py's cunet leg renders the 1080p frame (6 x 10 = 60 tiles) in ONE minibatch (tile batch 66) (the conv dispatch is shape-dependent: the
    DMA conv takes launches of more than 512 patches); the frame must carry the bits of the 16- and 7-tile minibatch renders, and
    an interior tile meets the oracle bound."""
    from nunif_amd.waifu2x.models.cunet import CUNet
    from nunif_amd.nunif.utils.render import tiled_render
    sd = OC.random_state_dict(202)
    m = CUNet().eval()
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    img = synth_image(63, 3, 1080, 1920)
    out = tiled_render(img, m, tile_size=256, batch_size=66)
    assert out.shape == (3, 1080, 1920)
    for mb in (16, 7):
        o = tiled_render(img, m, tile_size=256, batch_size=mb)
        assert torch.equal(out, o), f"minibatch 66 != minibatch {mb}: {float((out - o).abs().max())}"
    cfg = OS.create_config(1080, 1920, 1, 28, 256, 0)
    xp = torch.nn.functional.pad(img[None], cfg["pad"], mode="replicate")[0]
    z = OC.model_forward(sd, xp[:, 400:656, 800:1056][None])[0]        # tile (2,4) -> output [400:600) x [800:1000)
    assert psnr(out[:, 400:600, 800:1000].cpu(), z) >= 50.0


@pytest.mark.gpu
def test_dma_staged_conv_is_bit_identical_to_the_register_staged_one(hiplib, monkeypatch):
    """conv3_dma_kernel (persistent, halo + weights by LDS-DMA; launches of more than 512 patches) against conv3_lds_kernel on the
    same engines: same MFMA order over k, so the outputs must be EQUAL — cunet / upcunet on a 12-tile minibatch of 256 x 256
    (VALID convs, Cin 32 / 64, LeakyReLU) and the depth net's DPT head on a 4 x 392 x 686 batch (zero padding, pre-activation
    ReLU, two residuals)."""
    from nunif_amd.waifu2x.models.cunet import CUNet, UpCUNet
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    from oracle import depth_anything_v2 as ODA
    x = torch.stack([synth_image(70 + i, 3, 256, 256) for i in range(12)]).to("cuda:0")
    for cls, up in ((CUNet, False), (UpCUNet, True)):
        m = cls().eval()
        m.load_state_dict(OC.random_state_dict(205, up=up), strict=True)
        m = m.to("cuda:0")
        monkeypatch.setenv("NUNIF_CONV3_DMA", "0")
        a = m(x).clone()
        monkeypatch.setenv("NUNIF_CONV3_DMA", "1")
        monkeypatch.setenv("NUNIF_CONV3_DMA_KSPLIT", "0")
        b = m(x).clone()
        assert torch.isfinite(b).all() and torch.equal(a, b), float((a - b).abs().max())
        # the 128 -> 64 convs in two K halves on the DMA conv: same products, the halves summed one after the other
        monkeypatch.setenv("NUNIF_CONV3_DMA_KSPLIT", "1")
        bk = m(x).clone()
        assert torch.isfinite(bk).all() and psnr(b, bk) >= 57.0, psnr(b, bk)
        # the fused UNetConv(3, 32, 64) stem (conv1 on the MFMA with fp16 weights) against the VALU first conv + separate 32 -> 64 conv
        monkeypatch.setenv("NUNIF_CUNET_STEM", "0")
        c = m(x).clone()
        monkeypatch.setenv("NUNIF_CUNET_STEM", "1")
        assert psnr(b, c) >= 57.0, psnr(b, c)
        # the 2 x 2 stride-2 convs as gather GEMMs against conv_kernel: same products, another summation order
        monkeypatch.setenv("NUNIF_CUNET_DOWN_GEMM", "0")
        d = m(x).clone()
        monkeypatch.setenv("NUNIF_CUNET_DOWN_GEMM", "1")
        assert psnr(b, d) >= 57.0, psnr(b, d)
        # the 64 -> 3 image heads in tap-scatter form (cunet_head.hip) against the conv form, everything else as it is now (K halves on):
        # the same 576 fp32 products per output, channels summed first — ~1e-6 on unet1's head, whose output then runs through unet2's
        # fp16 layers; its two tile shapes compute every output in the same order
        b2 = m(x).clone()
        monkeypatch.setenv("NUNIF_CUNET_HEAD", "0")
        e = m(x).clone()
        monkeypatch.setenv("NUNIF_CUNET_HEAD", "1")
        assert psnr(b2, e) >= 59.0 and float((b2 - e).abs().max()) < 2e-3, (psnr(b2, e), float((b2 - e).abs().max()))   # (psnr() saturates at 60 dB)
        monkeypatch.setenv("NUNIF_CUNET_HEAD_TW", "32")
        f32 = m(x).clone()
        monkeypatch.delenv("NUNIF_CUNET_HEAD_TW")
        assert torch.equal(b2, f32), float((b2 - f32).abs().max())
    net = HipDepthAnythingV2(ODA.random_state_dict(601), "cuda:0")
    xd = torch.stack([synth_image(90 + i, 3, 392, 686) for i in range(4)]).to("cuda:0") * 2 - 1
    monkeypatch.setenv("NUNIF_CONV3_DMA", "0")
    a = net(xd).clone()
    monkeypatch.setenv("NUNIF_CONV3_DMA", "1")
    b = net(xd).clone()
    assert torch.isfinite(b).all() and torch.equal(a, b), float((a - b).abs().max())


def test_tap_scatter_form_of_the_image_head_is_the_valid_3x3_conv():
    """``cunet_head_kernel`` (nunif_amd/csrc/cunet_head.hip) in torch: contract the 64 channels first — T[p][3 tap + c] =
    sum_ci W[c][ci][tap] x[p][ci], one 64 -> 27 Linear per INPUT pixel — then out[y][x][c] = bias[c] + sum_tap T[(y + dy, x + dx)][3 tap + c].
    The same 576 products per output as Conv2d(64, 3, 3) VALID (waifu2x/models/cunet.py:62,120), summed in another order."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 21, 19, generator=g)
    w = torch.randn(3, 64, 3, 3, generator=g) * 0.1
    b = torch.randn(3, generator=g)
    ref = torch.nn.functional.conv2d(x, w, b)
    w27 = w.permute(2, 3, 0, 1).reshape(27, 64)                       # row n = 3 tap + c, tap = 3 dy + dx
    t = torch.einsum("nk,bkhw->bnhw", w27, x)                         # [B, 27, H, W]
    out = torch.zeros_like(ref)
    for tap in range(9):
        dy, dx = divmod(tap, 3)
        out += t[:, 3 * tap:3 * tap + 3, dy:dy + ref.shape[2], dx:dx + ref.shape[3]]
    out += b.view(1, 3, 1, 1)
    assert float((out - ref).abs().max()) < 1e-4
