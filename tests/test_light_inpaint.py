"""inpaint.light_inpaint_v1 + the image-mode MLBW inpaint flow: oracle vs the reference fixture (CPU), HIP engine vs fixture (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr, sd_checksum
from oracle import light_inpaint as OL
from oracle import mlbw as OM


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "light_inpaint.npz")).items()}


def test_oracle_matches_reference_fixture(g):
    sd = OL.random_state_dict(701)
    assert sd_checksum(sd) == pytest.approx(float(g["sdsum"]), rel=1e-12)
    x, mask = g["x"], g["mask"]
    assert (OL.infer(sd, x, mask) - g["infer"]).abs().max().item() < 1e-5
    assert (OL.infer(sd, x, mask, closing=True, inner_dilation=1, outer_dilation=2, base_width=50) - g["infer_close"]).abs().max().item() < 1e-5
    assert (OL.forward(sd, x, mask.float(), skip_i2i_offset=False) - g["forward_off"]).abs().max().item() < 1e-5
    ref = g["infer"]
    assert 0.05 < ref.std().item() < 0.4 and (ref - x).abs().mean().item() > 0.01        # the net really paints


def test_oracle_mlbw_inpaint_image_flow(g):
    sdm, sdi = OM.random_state_dict(431, 2, False, hole_mask=True), OL.random_state_dict(701)
    left, right = OL.mlbw_inpaint_image(sdm, sdi, g["c"], g["depth"], 2.0, 0.5, "both", 1, 1)
    assert (left - g["mi_left"].float()).abs().max().item() < 2e-3 and (right - g["mi_right"].float()).abs().max().item() < 2e-3
    le, ri = OL.mlbw_inpaint_image(sdm, sdi, g["c"][:1], g["depth"][:1], 2.0, 0.5, "right", 0, 0)
    assert le is not None and (ri - g["mi_right_only"].float()).abs().max().item() < 2e-3


def _models():
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    mi = create_model("inpaint.light_inpaint_v1").eval()
    mi.load_state_dict(OL.random_state_dict(701), strict=True)
    mm = create_model("sbs.mask_mlbw_l2").eval()
    mm.load_state_dict(OM.random_state_dict(431, 2, False, hole_mask=True), strict=True)
    return mi.to("cuda:0"), mm.to("cuda:0")


@pytest.mark.gpu
def test_hip_light_inpaint(hiplib, g):
    mi, _ = _models()
    assert (mi.name, mi.i2i_scale, mi.i2i_offset, mi.i2i_blend_size) == ("inpaint.light_inpaint_v1", 1, 16, 8)
    x, mask = g["x"].to("cuda:0"), g["mask"].to("cuda:0")
    y = mi.infer(x, mask)
    assert y.shape == x.shape and y.dtype == x.dtype
    p = psnr(y.cpu(), g["infer"])
    assert p >= 50.0, p
    keep = ~OL.preprocess(g["x"], g["mask"])[1].expand_as(g["x"]).gt(0)                  # outside the soft mask: the input
    assert torch.equal(y.cpu()[keep], g["x"][keep])
    y2 = mi.infer(x, mask, closing=True, inner_dilation=1, outer_dilation=2, base_width=50)
    p = psnr(y2.cpu(), g["infer_close"])
    assert p >= 50.0, p
    assert torch.equal(mi.infer(x[:1], mask[:1]), y[:1])                                  # batch invariance
    with pytest.raises(NotImplementedError):
        mi(x, mask.float())
    from nunif_amd.nunif.models import create_model
    with pytest.raises(RuntimeError):
        create_model("inpaint.light_inpaint_v1").eval().infer(g["x"], g["mask"])          # CPU-resident model: no fallback


@pytest.mark.gpu
def test_mirrored_infer_equals_flip_infer_flip_bit_for_bit(hiplib, g, gv):
    """``infer(x, mask, mirror_x=True)`` (``nunif_hip_light_inpaint_infer_ex``: the picture read and written at column W - 1 - x)
    == ``flip(infer(flip(x), mask))`` with the mask given in the flipped frame — what the side-model driver feeds the left eye
    (iw3/mlbw_inpaint.py:60-76) — for the image net and the 12-frame video net; and the driver's two routes give the same eyes."""
    from nunif_amd.nunif.models import create_model
    mi, _ = _models()
    x, mask = g["x"].to("cuda:0"), g["mask"].to("cuda:0")
    a = mi.infer(x.flip(-1).contiguous(), mask, closing=True, inner_dilation=1, outer_dilation=2, base_width=50).flip(-1)
    b = mi.infer(x, mask, closing=True, inner_dilation=1, outer_dilation=2, base_width=50, mirror_x=True)
    assert torch.equal(a, b) and not torch.equal(b, mi.infer(x, mask, closing=True, inner_dilation=1, outer_dilation=2, base_width=50))
    mv = create_model("inpaint.light_video_inpaint_v1").eval()
    mv.load_state_dict(OL.video_random_state_dict(702), strict=True)
    mv = mv.to("cuda:0")
    xv, mk = gv["x"].to("cuda:0"), gv["mask"].to("cuda:0")
    assert torch.equal(mv.infer(xv.flip(-1).contiguous(), mk).flip(-1), mv.infer(xv, mk, mirror_x=True))


@pytest.mark.gpu
def test_hip_mlbw_inpaint_image(hiplib, g):
    from nunif_amd.iw3.mlbw_inpaint import MLBWInpaint
    mi, mm = _models()
    side = MLBWInpaint(mi, mm)
    side.set_mode("image")
    c, depth = g["c"].to("cuda:0"), g["depth"].to("cuda:0")
    left, right = side.infer(c, depth, divergence=2.0, convergence=0.5, synthetic_view="both", inner_dilation=1, outer_dilation=1)
    for got, key in ((left, "mi_left"), (right, "mi_right")):
        ref = g[key].float()
        bad = ((got.cpu() - ref).abs() > 2e-2).float().mean().item()
        assert bad < 0.03, (key, bad)          # hole pixels whose logit sits within fp16 noise of the threshold may flip
    le, ri = side.infer(c[:1], depth[:1], divergence=2.0, convergence=0.5, synthetic_view="right")
    assert le is c[:1] or torch.equal(le, c[:1])
    assert ((ri.cpu() - g["mi_right_only"].float()).abs() > 2e-2).float().mean().item() < 0.03
    assert side.flush() == (None, None)
    with pytest.raises(NotImplementedError):
        side.set_mode("video")


# ---- inpaint.light_video_inpaint_v1 + MLBWInpaintVideo --------------------------------------------------------------------
@pytest.fixture(scope="module")
def gv():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "light_video_inpaint.npz")).items()}


def _video_batches(gv):
    f, d = gv["v_frames"], gv["v_depth"]
    return [(f[i:i + 3], d[i:i + 3]) for i in range(0, f.shape[0], 3)]


def test_oracle_video_matches_reference_fixture(gv):
    sd = OL.video_random_state_dict(801)
    assert sd_checksum(sd) == pytest.approx(float(gv["sdsum"]), rel=1e-12)
    assert (OL.video_infer(sd, gv["x"], gv["mask"]) - gv["infer12"]).abs().max().item() < 2e-5
    y7 = OL.video_infer(sd, gv["x"][:7], gv["mask"][:7], True, 1, 1, 36)
    assert y7.shape[0] == 7 and (y7 - gv["infer7"]).abs().max().item() < 2e-5


def test_oracle_video_queue_flow(gv):
    sdm, sdv = OM.random_state_dict(431, 2, False, hole_mask=True), OL.video_random_state_dict(801)
    res = OL.mlbw_inpaint_video(sdm, sdv, _video_batches(gv)[:3], 2.0, 0.5, 1, 1)        # first output + flush (CPU time)
    sizes = [0 if r is None else r[0].shape[0] for r in res]
    assert sizes == [0, 0, 6, 3]
    left = torch.cat([r[0] for r in res if r is not None])
    assert (left[:6] - gv["v_left"][:6].float()).abs().max().item() < 2e-3             # fixture stored as fp16


@pytest.mark.gpu
def test_hip_light_video_inpaint(hiplib, gv):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    mv = create_model("inpaint.light_video_inpaint_v1").eval()
    mv.load_state_dict(OL.video_random_state_dict(801), strict=True)
    mv = mv.to("cuda:0")
    x, mask = gv["x"].to("cuda:0"), gv["mask"].to("cuda:0")
    y = mv.infer(x, mask)
    p = psnr(y.cpu(), gv["infer12"])
    assert y.shape == x.shape and p >= 50.0, p
    y7 = mv.infer(x[:7], mask[:7], closing=True, inner_dilation=1, outer_dilation=1, base_width=36)
    p = psnr(y7.cpu(), gv["infer7"])
    assert y7.shape[0] == 7 and p >= 50.0, p
    with pytest.raises(AssertionError):
        mv.infer(torch.cat([x, x[:1]]), torch.cat([mask, mask[:1]]))


@pytest.mark.gpu
def test_large_frame_kernel_choices_at_small_size(hiplib, g, gv, monkeypatch):
    """At 4K the inpaint nets' K = 192 Linears run on the resident-weight GEMM (and the 192 -> 768 one as two launches over the
    output halves), chosen by token count (>= 2^20: light_inpaint.hip lin(), swin_kernels.hip gemm_big_m).  The fixtures are far
    smaller, so the threshold is lowered to 1 here: same fixtures, same PSNR bar, and the two choices agree to fp16 rounding."""
    mi, _ = _models()
    x, mask = g["x"].to("cuda:0"), g["mask"].to("cuda:0")
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    mv = create_model("inpaint.light_video_inpaint_v1").eval()
    mv.load_state_dict(OL.video_random_state_dict(801), strict=True)
    mv = mv.to("cuda:0")
    xv, mkv = gv["x"].to("cuda:0"), gv["mask"].to("cuda:0")
    a, av = mi.infer(x, mask).clone(), mv.infer(xv, mkv).clone()
    monkeypatch.setenv("NUNIF_GEMM_BIG_M", "1")
    b, bv = mi.infer(x, mask).clone(), mv.infer(xv, mkv).clone()
    assert psnr(b.cpu(), g["infer"]) >= 50.0 and psnr(bv.cpu(), gv["infer12"]) >= 50.0
    assert psnr(a, b) >= 60.0 and psnr(av, bv) >= 60.0, (psnr(a, b), psnr(av, bv))


@pytest.mark.gpu
def test_hip_mlbw_inpaint_video_queue(hiplib, gv):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3.mlbw_inpaint import MLBWInpaint
    mi, mm = _models()
    mv = create_model("inpaint.light_video_inpaint_v1").eval()
    mv.load_state_dict(OL.video_random_state_dict(801), strict=True)
    side = MLBWInpaint(mi, mm, video_model=mv.to("cuda:0"))
    side.set_mode("video")
    sizes, lefts, rights = [], [], []
    for f, d in _video_batches(gv):
        le, ri = side.infer(f.to("cuda:0"), d.to("cuda:0"), divergence=2.0, convergence=0.5, synthetic_view="both",
                            inner_dilation=1, outer_dilation=1)
        sizes.append(0 if le is None else le.shape[0])
        if le is not None:
            lefts.append(le.clone()); rights.append(ri.clone())
    le, ri = side.flush()
    sizes.append(0 if le is None else le.shape[0])
    lefts.append(le); rights.append(ri)
    assert sizes == [int(v) for v in gv["v_sizes"]] == [0, 0, 6, 0, 6, 0, 6]
    for got, key in ((torch.cat(lefts), "v_left"), (torch.cat(rights), "v_right")):
        ref = gv[key].float()
        assert got.shape == ref.shape
        bad = ((got.cpu() - ref).abs() > 2e-2).float().mean().item()
        assert bad < 0.03, (key, bad)
    assert side.flush() == (None, None)
    side.reset()


# ---- inpaint.light_video_inpaint_v1_medium / _large (base_dim 128 / 192, lv2_mlp_ratio 2) -----------------------------------
VARIANTS = (("medium", "inpaint.light_video_inpaint_v1_medium", 811, 128), ("large", "inpaint.light_video_inpaint_v1_large", 812, 192))


def _gml():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "light_video_inpaint_ml.npz")).items()}


def test_oracle_video_medium_large_match_reference_fixture():
    g = _gml()
    x, mask = g["x"].float(), g["mask"]
    for tag, _, seed, dim in VARIANTS:
        sd = OL.video_random_state_dict(seed, base_dim=dim, lv2_mlp_ratio=2)
        assert sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point}) == pytest.approx(float(g["sdsum_" + tag]),
                                                                                                        rel=1e-12)
        y = OL.video_infer(sd, x, mask, inner_dilation=1)
        assert (y - g["y_" + tag].float()).abs().max().item() < 1e-3, tag        # fixture stored as fp16


@pytest.mark.gpu
def test_hip_light_video_inpaint_medium_large(hiplib):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    g = _gml()
    x, mask = g["x"].float().to("cuda:0"), g["mask"].to("cuda:0")
    for tag, name, seed, dim in VARIANTS:
        mv = create_model(name).eval()
        mv.load_state_dict(OL.video_random_state_dict(seed, base_dim=dim, lv2_mlp_ratio=2), strict=True)
        y = mv.to("cuda:0").infer(x, mask, inner_dilation=1)
        p = psnr(y.cpu(), g["y_" + tag].float())
        assert y.shape == x.shape and p >= 50.0, (tag, p)
    with pytest.raises(ValueError):
        create_model("inpaint.light_video_inpaint_v1", base_dim=64)
