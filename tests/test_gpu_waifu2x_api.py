"""Waifu2x context / hub API on the HIP engine, end to end through ``.pth`` files (contract B3/B4)."""
import os

import pytest
import torch

from conftest import psnr, synth_image
from oracle import alpha_tta as OA
from oracle import cunet as OC
from oracle import seam_blending as OS
from oracle import swin_unet as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory, hiplib):
    """pretrained_models/swin_unet/art with seeded weights, written in the reference's .pth container."""
    from nunif_amd.nunif.models import save_model
    from nunif_amd.waifu2x.models import swin_unet as M
    root = tmp_path_factory.mktemp("models")
    d = root / "swin_unet" / "art"
    d.mkdir(parents=True)
    sds = {}
    for fname, cls, sf, seed in (("scale2x.pth", M.SwinUNet2x, 2, 301), ("noise1_scale2x.pth", M.SwinUNet2x, 2, 302),
                                 ("noise1.pth", M.SwinUNet, 1, 303), ("scale4x.pth", M.SwinUNet4x, 4, 304)):
        sd = O.random_state_dict(seed, sf)
        m = cls()
        m.load_state_dict(sd)
        save_model(m, str(d / fname))
        sds[fname] = sd
    return str(root), str(d), sds


def oracle_render(sd, name, tile=64):
    s, off, blend, _ = O.GEOMETRY[name]
    return lambda x: OS.tiled_render(x, lambda mb: O.model_forward(sd, mb, name), s, off, blend, tile, 4)


def test_convert_scale_noise_and_errors(model_dir):
    from nunif_amd.waifu2x.utils import Waifu2x
    _, d, sds = model_dir
    ctx = Waifu2x(d, [0])
    ctx.load_model("scale", -1)
    ctx.load_model("noise_scale", 1)
    ctx.load_model("noise", 1)
    x = synth_image(71, 3, 70, 90)
    with torch.inference_mode():
        rgb, alpha = ctx.convert(x, None, "scale", -1, tile_size=64, batch_size=4)
        assert alpha is None and rgb.device.type == "cpu" and rgb.shape == (3, 140, 180)
        assert psnr(rgb, oracle_render(sds["scale2x.pth"], "waifu2x.swin_unet_2x")(x)) >= 50.0
        rgb, _ = ctx.convert(x, None, "noise_scale", 1, tile_size=64, batch_size=4, output_device="cuda:0")
        assert rgb.device.type == "cuda"
        assert psnr(rgb.cpu(), oracle_render(sds["noise1_scale2x.pth"], "waifu2x.swin_unet_2x")(x)) >= 50.0
        rgb, _ = ctx.convert(x, None, "noise", 1, tile_size=64)
        assert rgb.shape == (3, 70, 90)
        assert psnr(rgb, oracle_render(sds["noise1.pth"], "waifu2x.swin_unet_1x")(x)) >= 50.0
        with pytest.raises(AssertionError):
            ctx.convert(x, None, "bogus", 0)
        with pytest.raises(AssertionError):
            ctx.convert(x[:2], None, "scale", -1)
    with pytest.raises(AssertionError):           # grad enabled
        with torch.enable_grad():
            ctx.convert(x, None, "scale", -1)
    with pytest.raises(FileNotFoundError):
        Waifu2x(d, [0]).load_model("noise_scale4x", 2)
    with pytest.raises(ValueError):
        Waifu2x(d, [0])._load_model("upscale", 0)


def test_convert_alpha_and_tta(model_dir):
    from nunif_amd.waifu2x.utils import Waifu2x
    _, d, sds = model_dir
    ctx = Waifu2x(d, [0])
    ctx.load_model("scale", -1)
    x = synth_image(72, 3, 50, 60)
    g = torch.Generator().manual_seed(5)
    alpha = ((torch.rand(1, 50, 60, generator=g) > 0.3).float() * torch.rand(1, 50, 60, generator=g)).contiguous()
    render = oracle_render(sds["scale2x.pth"], "waifu2x.swin_unet_2x")
    with torch.inference_mode():
        rgb, a = ctx.convert(x, alpha, "scale", -1, tile_size=64, batch_size=4)
        ref_rgb, ref_a = OA.convert(render, x, alpha, 2, 16)
        assert psnr(rgb, ref_rgb) >= 50.0 and a.shape == (1, 100, 120) and psnr(a, ref_a) >= 50.0
        rgb, a = ctx.convert(x, torch.ones(1, 50, 60), "scale", -1, tile_size=64)      # blank alpha: nearest x2
        assert torch.equal(a, torch.ones(1, 100, 120))
        rgb, _ = ctx.convert(x, None, "scale", -1, tile_size=64, tta=True)
        ref, _ = OA.convert(render, x, None, 2, 16, tta=True)
        assert psnr(rgb, ref) >= 50.0


def test_scale_falls_back_to_downscaled_4x(model_dir, tmp_path):
    """No scale2x.pth in the directory -> `scale` = scale4x.to_2x() (waifu2x/utils.py:139-144)."""
    import shutil
    from nunif_amd.waifu2x.utils import Waifu2x
    _, d, sds = model_dir
    shutil.copy(os.path.join(d, "scale4x.pth"), tmp_path / "scale4x.pth")
    ctx = Waifu2x(str(tmp_path), [0])
    ctx.load_model("scale", -1)
    assert ctx.scale_model.name == "waifu2x.swin_unet_downscaled" and ctx.scale_model.i2i_scale == 2
    x = synth_image(73, 3, 64, 64)
    with torch.inference_mode():
        rgb, _ = ctx.convert(x, None, "scale", -1, tile_size=64)
    sd = sds["scale4x.pth"]
    ref = OS.tiled_render(x, lambda mb: O.model_forward(sd, mb, downscale_factor=2), 2, 16, 8, 64, 4)
    assert psnr(rgb, ref) >= 50.0


def test_hub_image_model(model_dir):
    from PIL import Image
    from nunif_amd.waifu2x.hub import Waifu2xImageModel, waifu2x
    root, d, sds = model_dir
    m = Waifu2xImageModel("art", method="scale", device_ids=[0], tile_size=64, batch_size=4, model_root=root)
    x = synth_image(74, 3, 40, 56)
    rgb, alpha = m.infer(x, output_type="tensor")
    assert alpha is None and psnr(rgb, oracle_render(sds["scale2x.pth"], "waifu2x.swin_unet_2x")(x)) >= 50.0
    pil = Image.fromarray((x.permute(1, 2, 0) * 255).round().byte().numpy(), "RGB")
    out = m(pil)
    assert out.size == (112, 80) and out.mode == "RGB"
    m.set_mode("scale", 1)                         # noise_level >= 0 turns scale into noise_scale (no loading)
    assert m.method == "noise_scale" and m.noise_level == 1
    m1 = Waifu2xImageModel("art", method="scale2x", noise_level=1, device_ids=[0], tile_size=64, model_root=root)
    assert m1.method == "noise_scale" and m1.ctx.scale_model is not None      # companion alpha model loaded too
    assert m1.infer(x, output_type="tensor")[0].shape == (3, 80, 112)
    with pytest.raises(ValueError):
        Waifu2xImageModel("no_such_type", model_root=root)
    with pytest.raises(ValueError):
        m.infer(12345)
    m2 = waifu2x("art", method="noise", noise_level=1, device_ids=[0], tile_size=64, model_root=root)
    assert m2.infer(x, output_type="tensor")[0].shape == (3, 40, 56)


def test_cunet_through_the_context(tmp_path, hiplib):
    """BASELINE config 0 plumbing: `waifu2x cunet noise1` on a 512x512 image, here on the engine."""
    from nunif_amd.nunif.models import save_model
    from nunif_amd.waifu2x.models.cunet import CUNet
    from nunif_amd.waifu2x.utils import Waifu2x
    sd = OC.random_state_dict(305)
    m = CUNet()
    m.load_state_dict(sd)
    save_model(m, str(tmp_path / "noise1.pth"))
    ctx = Waifu2x(str(tmp_path), [0])
    ctx.load_model("noise", 1)
    x = synth_image(75, 3, 128, 128)
    with torch.inference_mode():
        rgb, _ = ctx.convert(x, None, "noise", 1, tile_size=96, batch_size=4)
    ref = OS.tiled_render(x, lambda mb: OC.model_forward(sd, mb), 1, 28, 0, 96, 4)
    assert rgb.shape == (3, 128, 128) and psnr(rgb, ref) >= 50.0


@pytest.mark.gpu
def test_hip_tta_views_and_merge_bit_exact(hiplib):
    from nunif_amd.nunif.transforms.tta import tta_merge, tta_split
    g = torch.Generator().manual_seed(11)
    x = torch.rand(3, 37, 53, generator=g)
    views = tta_split(x.to("cuda:0"))
    ref_views = OA.tta_split(x)
    assert len(views) == 8
    for v, r in zip(views, ref_views):
        assert v.shape == r.shape and torch.equal(v.cpu(), r)
    # merge of arbitrary per-view "model outputs" (different per view), reference order of additions
    outs = [torch.rand(r.shape, generator=g) for r in ref_views]
    import nunif_amd.nunif.transforms.tta as T
    got = tta_merge([o.to("cuda:0") for o in outs]).cpu()
    avg = outs[0].clone()
    for k, y in enumerate(outs[1:], start=1):
        if k & 1:
            y = torch.flip(y, (2,))
        if k & 2:
            y = torch.flip(y, (1,))
        if k & 4:
            y = torch.rot90(y, -1, (1, 2))
        avg += y
    avg *= 1 / 8.0
    assert torch.equal(got, torch.clamp(avg, 0, 1))
    assert T.tta_merge.__module__.startswith("nunif_amd")
    with pytest.raises(RuntimeError):
        tta_split(x)                                        # CPU tensor: no fallback


@pytest.mark.gpu
def test_hip_alpha_border_padding(hiplib):
    from nunif_amd.nunif.utils.alpha import AlphaBorderPadding
    g = torch.Generator().manual_seed(12)
    rgb = torch.rand(3, 64, 80, generator=g)
    alpha = ((torch.rand(1, 64, 80, generator=g) > 0.6).float() * torch.rand(1, 64, 80, generator=g)).contiguous()
    alpha[:, 20:44, 30:60] = 0.0                              # a hole wider than the padding radius
    pad = AlphaBorderPadding()
    for offset in (0, 1, 4, 16):
        got = pad(rgb.to("cuda:0"), alpha.to("cuda:0"), offset).cpu()
        ref = OA.alpha_border_padding(rgb, alpha, offset)
        assert (got - ref).abs().max().item() < 1e-5, offset
    # pixels deeper than `offset` inside the hole stay zero
    assert float(pad(rgb.to("cuda:0"), alpha.to("cuda:0"), 4)[:, 30:34, 42:48].abs().max()) == 0.0


@pytest.mark.gpu
def test_concurrent_renderer_matches_single_stream(hiplib):
    """nunif_amd.parallel.ConcurrentRenderer: frames dealt over 2 / 3 engine replicas on their own HIP streams come back in
    submission order and bit-identical to the single-stream render (results do not depend on what shares the GPU)."""
    from nunif_amd.nunif.utils.render import tiled_render
    from nunif_amd.parallel import ConcurrentRenderer
    from nunif_amd.synthetic import swin_unet_state_dict
    from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x
    sd = swin_unet_state_dict(102, 2)

    def make():
        m = SwinUNet2x().eval()
        m.load_state_dict(sd)
        return m

    g = torch.Generator().manual_seed(5)
    frames = [torch.rand(3, 150 + 10 * i, 200, generator=g).to("cuda:0") for i in range(7)]

    def render(m, f):
        return tiled_render(f, m, tile_size=64, batch_size=4)

    ref_model = make().to("cuda:0")
    ref = [render(ref_model, f).cpu() for f in frames]
    for n in (2, 3):
        pool = ConcurrentRenderer(make, n, "cuda:0")
        outs = [o.cpu() for o in pool.map(render, frames)]
        assert len(outs) == len(ref)
        for a, b in zip(outs, ref):
            assert a.shape == b.shape and torch.equal(a, b)
        pool.synchronize()
    with pytest.raises(RuntimeError):
        ConcurrentRenderer(make, 2, "cpu")
