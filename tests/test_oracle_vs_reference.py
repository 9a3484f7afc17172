"""Cross-check of the oracle against the LIVE reference (build container only; skipped where /root/reference is
absent, e.g. on the GPU box — the committed golden fixtures cover that case)."""
import pytest
import torch

from oracle import refstub
from oracle import seam_blending as OS
from oracle import swin_unet as O

pytestmark = pytest.mark.skipif(not refstub.reference_available(), reason="reference tree not mounted")


@pytest.fixture(scope="module")
def ref():
    refstub.install()
    import waifu2x.models.swin_unet as ref_swin
    import nunif.utils.seam_blending as ref_seam
    return {"swin": ref_swin, "seam": ref_seam}


def test_param_counts_and_keys(ref):
    for cls, sf, n in ((ref["swin"].SwinUNet, 1, 3757431), (ref["swin"].SwinUNet2x, 2, 3758304),
                       (ref["swin"].SwinUNet4x, 4, 4302852)):
        m = cls()
        assert sum(p.numel() for p in m.parameters()) == n
        sd = O.random_state_dict(7, sf)
        msd = m.state_dict()
        assert set(msd) == set(sd)
        assert all(msd[k].shape == sd[k].shape for k in sd)
        assert torch.equal(msd["unet.swin1.block.1.attn.relative_position_index"],
                           sd["unet.swin1.block.1.attn.relative_position_index"])


def test_random_sizes_grid(ref):
    g = torch.Generator().manual_seed(5)
    SB = ref["seam"].SeamBlending
    for _ in range(200):
        h, w = (int(v) for v in torch.randint(1, 3000, (2,), generator=g))
        s, o, b, t = [(1, 8, 4, 64), (2, 16, 8, 256), (4, 32, 16, 112), (1, 28, 0, 256), (2, 36, 0, 128)][
            int(torch.randint(0, 5, (1,), generator=g))]
        assert SB.create_config((h, w), s, o, t, b) == OS.create_config(h, w, s, o, t, b)


def test_forward_2x_random_tile(ref):
    sd = O.random_state_dict(31, 2)
    m = ref["swin"].SwinUNet2x().eval()
    m.load_state_dict(sd)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    assert (m(x) - O.model_forward(sd, x)).abs().max().item() < 2e-4


@pytest.mark.parametrize("view", ["both", "left", "right"])
def test_stereo_window_matches_the_reference_frame_queue_under_random_operations(view):
    """The 12-frame inpaint window: the product's ``StereoWindow`` (side_model.py — own API: push / pad_with_last / slide / reset)
    against the reference's ``FrameQueue`` (iw3/inpaint_utils.py:98-188: add / fill / remove / clear), contents and fill level
    state for state under the operation pattern the video drivers produce."""
    refstub.install()
    import av
    av.__version__ = "14.2.0"
    from iw3.inpaint_utils import FrameQueue as RefQueue
    from nunif_amd.iw3.side_model import StereoWindow
    a = RefQueue(synthetic_view=view, seq=12, height=6, width=10, dtype=torch.float32, device="cpu", mask_height=3, mask_width=5)
    b = StereoWindow(view, 12, (3, 6, 10), (1, 3, 5), torch.float32, "cpu")
    g = torch.Generator().manual_seed(17)
    names = {"both": ("left", "right", "left_mask", "right_mask"), "left": ("left", "right", "left_mask"),
             "right": ("left", "right", "right_mask")}[view]

    def same():
        assert (a.index, a.full(), a.empty()) == (b.level, b.is_full(), b.is_empty())
        for x, name in zip(a.get(), names):
            assert torch.equal(x[:a.index], b.get(name)[:b.level])

    for step in range(60):
        if a.full():
            if int(torch.randint(0, 3, (1,), generator=g)) == 1:
                a.clear(); b.reset()
            else:
                a.remove(6); b.slide(6)
        elif not a.empty() and int(torch.randint(0, 6, (1,), generator=g)) == 0:
            assert a.fill() == b.pad_with_last()
        else:
            le, ri = torch.rand(3, 6, 10, generator=g), torch.rand(3, 6, 10, generator=g)
            lm = torch.rand(1, 3, 5, generator=g) if view in ("both", "left") else None
            rm = torch.rand(1, 3, 5, generator=g) if view in ("both", "right") else None
            a.add(le, ri, **{k: v for k, v in (("left_mask", lm), ("right_mask", rm)) if v is not None})
            b.push(left=le, right=ri, left_mask=lm, right_mask=rm)
        same()


@pytest.mark.parametrize("decay,buffer_size,mode", [(0.0, 1, "minmax"), (0.75, 4, "minmax"), (0.9, 2, "max"), (0.5, 22, "minmax")])
def test_ema_minmax_scaler_matches_reference(decay, buffer_size, mode):
    """EMAMinMaxScaler / MinMaxBuffer (iw3/depth_scaler.py:33-142): the product class against the reference class over a frame
    sequence with flushes at scene cuts, output for output (both are torch-CPU here; on the device the same tensor ops run)."""
    refstub.install()
    from iw3.depth_scaler import EMAMinMaxScaler as RefScaler
    from nunif_amd.iw3.depth_scaler import EMAMinMaxScaler
    a, b = RefScaler(decay=decay, buffer_size=buffer_size, mode=mode), EMAMinMaxScaler(decay=decay, buffer_size=buffer_size, mode=mode)
    g = torch.Generator().manual_seed(23)
    n_out = 0
    for i in range(70):
        frame = torch.rand(1, 5, 7, generator=g) * (0.5 + 0.05 * (i % 9)) + 0.01 * i
        ra, rb = a.update(frame, return_minmax=True), b.update(frame, return_minmax=True)
        assert (ra[0] is None) == (rb[0] is None)
        if ra[0] is not None:
            n_out += 1
            assert torch.equal(ra[0], rb[0]) and float(ra[1]) == float(rb[1]) and float(ra[2]) == float(rb[2])
        if i in (24, 25, 60):                                   # scene cuts (one right after another, one late)
            fa, fb = a.flush(), b.flush()
            assert len(fa) == len(fb) and all(torch.equal(x, y) for x, y in zip(fa, fb))
            n_out += len(fa)
    fa, fb = a.flush(return_minmax=True), b.flush(return_minmax=True)
    assert len(fa) == len(fb) and all(torch.equal(x[0], y[0]) for x, y in zip(fa, fb))
    assert n_out + len(fa) == 70


def test_host_tables_match_reference():
    """Pure host logic compared with the reference functions over their whole argument range: mapper name resolution
    (iw3/mapper.py:154-232), MLBW divergence levels (stereo_model_factory.py:36-42), 16-bit pixel formats (video.py:272-279),
    the depth pre-processing size rule (depth_anything_model.py:69-100)."""
    refstub.install()
    import av
    av.__version__ = "14.2.0"
    import iw3.mapper as RM
    import iw3.stereo_model_factory as RS
    import nunif.utils.video as RVU
    from nunif_amd.iw3 import mapper as M
    from nunif_amd.iw3 import stereo_model_factory as S
    from nunif_amd.iw3.depth_anything_model import preprocess_size
    from nunif_amd.iw3.frame_pipeline import pix_fmt_requires_16bit
    for metric in (False, True):
        for mtype in ((None, "div") if metric else (None, "mul", "shift")):
            assert M.get_mapper_levels(metric, mtype) == RM.get_mapper_levels(metric, mtype)
            for k in range(-12, 13):
                fs = k / 4.0
                assert M.resolve_mapper_name(None, fs, metric, mtype) == RM.resolve_mapper_name(None, fs, metric, mtype), (fs, metric, mtype)
        assert M.resolve_mapper_name("auto", 0, metric) == RM.resolve_mapper_name("auto", 0, metric)
    for d in [x / 4.0 for x in range(0, 50)]:
        assert S.get_mlbw_divergence_level(d) == RS.get_mlbw_divergence_level(d)
    for fmt in ("yuv420p", "yuv420p10le", "p010le", "yuv444p16le", "gbrp12le", "rgb48le", "rgb24", None):
        assert pix_fmt_requires_16bit(fmt) == RVU.pix_fmt_requires_16bit(fmt)
    # size rule: the reference computes it inside batch_preprocess; run it on zero frames and read the shape back
    from iw3.depth_anything_model import batch_preprocess
    g = torch.Generator().manual_seed(31)
    for _ in range(60):
        h, w = (int(v) for v in torch.randint(40, 900, (2,), generator=g))
        lb = int(torch.randint(16, 40, (1,), generator=g)) * 14
        for limit in (False, True):
            out = batch_preprocess(torch.zeros(1, 3, h, w), lower_bound=lb, limit_resolution=limit)
            assert tuple(out.shape[-2:]) == tuple(preprocess_size(h, w, lb, limit_resolution=limit)), (h, w, lb, limit)


def test_postprocess_padding_matches_reference():
    """``postprocess_padding`` (iw3/utils.py:394-427) is integer arithmetic + zero padding: product vs reference over random sizes."""
    refstub.install()
    import av
    av.__version__ = "14.2.0"
    import iw3.utils as RU
    from nunif_amd.iw3.utils import postprocess_padding

    class TFShim:                      # torchvision's tensor pad: (left, top, right, bottom)
        @staticmethod
        def pad(img, padding, padding_mode="constant"):
            le, t, r, b = padding
            return torch.nn.functional.pad(img, (le, r, t, b), mode=padding_mode)
    RU.TF = TFShim
    g = torch.Generator().manual_seed(41)
    for _ in range(80):
        h, w = (int(v) for v in torch.randint(8, 200, (2,), generator=g))
        pad = float(torch.rand(1, generator=g)) * 0.4
        mode = ["tblr", "tb", "lr", "top", "16:9"][int(torch.randint(0, 5, (1,), generator=g))]
        le, ri = torch.rand(3, h, w, generator=g), torch.rand(3, h, w, generator=g)
        a = RU.postprocess_padding(le, ri, pad, mode)
        b = postprocess_padding(le, ri, pad, mode)
        assert a[0].shape == b[0].shape and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (h, w, pad, mode)


@pytest.mark.parametrize("name", ["waifu2x.cunet", "waifu2x.upcunet", "waifu2x.vgg_7", "waifu2x.upconv_7", "waifu2x.swin_unet_1x",
                                  "waifu2x.swin_unet_2x", "waifu2x.swin_unet_4x", "waifu2x.swin_unet_8x", "waifu2x.swin_unet_4xl",
                                  "sbs.row_flow_v3", "sbs.mlbw_l2", "sbs.mlbw_l4", "sbs.mlbw_l2s", "sbs.mlbw_l4s", "sbs.mask_mlbw_l2",
                                  "iw3.depth_aa", "inpaint.light_inpaint_v1", "inpaint.light_video_inpaint_v1",
                                  "inpaint.light_video_inpaint_v1_medium", "inpaint.light_video_inpaint_v1_large"])
def test_pth_container_round_trips_between_the_two_trees(name, tmp_path):
    """The ``.pth`` container (nunif/models/utils.py:15-74) is the drop-in boundary for weights: a file written by the reference
    loads into the engine's class of the same registered name and a file written by the engine loads into the reference's, key for
    key."""
    refstub.install()
    import av
    av.__version__ = "14.2.0"
    import nunif.models as RNM
    import waifu2x.models  # noqa: F401  (registers the reference's waifu2x.*)
    import iw3.models  # noqa: F401
    from nunif_amd.nunif import models as PNM
    import nunif_amd.waifu2x.models.cunet, nunif_amd.waifu2x.models.swin_unet, nunif_amd.iw3.models  # noqa: F401,E401
    import nunif_amd.waifu2x.models.vgg_7, nunif_amd.waifu2x.models.upconv_7  # noqa: F401,E401
    torch.manual_seed(5)
    ref_model = RNM.create_model(name).eval()
    p1 = str(tmp_path / "from_reference.pth")
    RNM.save_model(ref_model, p1)
    ours, meta = PNM.load_model(p1, weights_only=True)
    assert ours.name == ref_model.name == meta["name"]        # (factory names such as sbs.mlbw_l2 build a model named sbs.mlbw)
    ref_sd = ref_model.state_dict()
    our_sd = ours.state_dict()
    assert set(our_sd) == set(ref_sd)
    assert all(torch.equal(our_sd[k].cpu(), ref_sd[k]) for k in ref_sd)
    assert (ours.i2i_scale, ours.i2i_offset) == (ref_model.i2i_scale, ref_model.i2i_offset)
    p2 = str(tmp_path / "from_engine.pth")
    PNM.save_model(ours, p2)
    back, _ = RNM.load_model(p2, weights_only=True)
    assert all(torch.equal(back.state_dict()[k], ref_sd[k]) for k in ref_sd)


def test_i2i_contract_attributes_match_for_every_registered_model():
    """Contract B1 (nunif/models/model.py:65-86): scale / offset / blend size / default tile and batch sizes and the tile-size
    validators (``find_valid_tile_size`` over 8 .. 1024) of every registered model, engine class vs reference class."""
    refstub.install()
    import av
    av.__version__ = "14.2.0"
    import nunif.models as RNM
    import waifu2x.models  # noqa: F401
    import iw3.models  # noqa: F401
    from nunif_amd.nunif import models as PNM
    import nunif_amd.waifu2x.models.cunet, nunif_amd.waifu2x.models.swin_unet, nunif_amd.iw3.models  # noqa: F401,E401
    import nunif_amd.waifu2x.models.vgg_7, nunif_amd.waifu2x.models.upconv_7  # noqa: F401,E401
    names = ["waifu2x.cunet", "waifu2x.upcunet", "waifu2x.vgg_7", "waifu2x.upconv_7", "waifu2x.swin_unet_1x", "waifu2x.swin_unet_2x",
             "waifu2x.swin_unet_4x", "waifu2x.swin_unet_8x", "waifu2x.swin_unet_4xl", "sbs.row_flow_v3", "sbs.mlbw_l2", "sbs.mask_mlbw_l2",
             "iw3.depth_aa", "inpaint.light_inpaint_v1", "inpaint.light_video_inpaint_v1"]
    for name in names:
        a, b = RNM.create_model(name), PNM.create_model(name)
        for attr in ("name", "i2i_scale", "i2i_offset", "i2i_in_channels", "i2i_blend_size", "i2i_default_tile_size",
                     "i2i_default_batch_size"):
            assert getattr(a, attr, None) == getattr(b, attr, None), (name, attr, getattr(a, attr, None), getattr(b, attr, None))
        def valid(m, t):
            try:
                return m.find_valid_tile_size(t)
            except ValueError:
                return "ValueError"              # below the smallest valid tile both raise
        for t in list(range(8, 1025, 4)) + [None]:
            assert valid(a, t) == valid(b, t), (name, t, valid(a, t), valid(b, t))


def test_hub_keep_alpha_false_composites_on_white_like_pil_io():
    """Waifu2xImageModel.infer_file(keep_alpha=False): the reference loads through pil_io._load_image(keep_alpha=False),
    which pastes the image onto a white background (nunif/utils/pil_io.py:26-30,:60-70) — pixels must match exactly."""
    refstub.install()
    import numpy as np
    from PIL import Image
    from nunif.utils import pil_io
    from nunif_amd.waifu2x.hub import _from_pil
    arr = np.random.RandomState(0).randint(0, 256, (8, 9, 4), dtype=np.uint8)
    im = Image.fromarray(arr, "RGBA")
    rgb, alpha = _from_pil(im, False)
    ref, _ = pil_io._load_image(im.copy(), "x.png", keep_alpha=False)
    assert alpha is None and ref.mode == "RGB"
    assert torch.equal((rgb * 255).round().to(torch.uint8), torch.from_numpy(np.asarray(ref).copy()).permute(2, 0, 1))
    rgb2, alpha2 = _from_pil(im, True)
    assert alpha2 is not None and torch.equal((rgb2 * 255).round().to(torch.uint8), torch.from_numpy(arr[..., :3].copy()).permute(2, 0, 1))


class _FakeI2I(torch.nn.Module):
    """A deterministic stand-in for an I2I net (fp32 CPU): nearest x``scale`` up-sampling of a per-pixel polynomial of the
    input with a position-dependent term, cropped by ``offset`` — every tile gets different values in its overlap region,
    so any deviation in the blend recurrence shows."""

    def __init__(self, scale, offset, blend_size):
        super().__init__()
        self.i2i_scale, self.i2i_offset, self.i2i_blend_size = scale, offset, blend_size
        self.i2i_default_batch_size, self.i2i_default_tile_size = 4, 64
        self.w = torch.nn.Parameter(torch.zeros(1))

    def find_valid_tile_size(self, t):
        return t

    def forward(self, x):
        s, o = self.i2i_scale, self.i2i_offset
        z = torch.nn.functional.interpolate(x, scale_factor=s, mode="nearest") if s > 1 else x
        ramp = torch.linspace(0, 0.3, z.shape[-1]).view(1, 1, 1, -1) + torch.linspace(0, 0.2, z.shape[-2]).view(1, 1, -1, 1)
        z = z * 0.7 + ramp * (0.5 + x.mean(dim=(1, 2, 3), keepdim=True))
        return z[:, :, o:z.shape[-2] - o, o:z.shape[-1] - o].contiguous()


@pytest.mark.parametrize("scale,offset,blend,tile,h,w,bs", [
    (2, 16, 8, 64, 100, 130, 4), (1, 8, 4, 64, 97, 64, 3), (4, 32, 16, 64, 70, 150, 5), (2, 16, 0, 64, 90, 90, 4),
    (2, 16, 8, 48, 48, 49, 2)])
def test_oracle_stitcher_is_bit_identical_to_reference_tiled_render(scale, offset, blend, tile, h, w, bs):
    """``oracle.seam_blending.tiled_render`` == the live ``SeamBlending.tiled_render`` (nunif/utils/seam_blending.py:48-106,
    update :156-174) with the SAME fake model: torch.equal — the seams are pinned bit for bit, not to a PSNR."""
    refstub.install()
    from nunif.utils.seam_blending import SeamBlending as RefSB
    from oracle import seam_blending as OS
    torch.manual_seed(scale * 1000 + h)
    x = torch.rand(3, h, w)
    model = _FakeI2I(scale, offset, blend).eval()
    with torch.inference_mode():
        ref = RefSB.tiled_render(x, model, tile_size=tile, batch_size=bs, enable_amp=False)
        got = OS.tiled_render(x, lambda mb: model(mb), scale, offset, blend, tile, bs)
    assert ref.shape == got.shape == (3, h * scale, w * scale)
    assert torch.equal(ref, got)
