"""iw3 output formats (postprocess_image: IPD / --pad modes, VR180, anaglyph, RGBD): oracle vs the reference fixture (CPU),
HIP engine vs fixture (GPU)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr
from oracle import iw3_utils as OU

sys.path.insert(0, GOLDEN)
from make_golden_cases import FORMAT_CASES  # noqa: E402


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "formats.npz")).items()}


def _oracle_kwargs(kw):
    kw = dict(kw)
    if "anaglyph" in kw:
        kw["anaglyph_kind"] = kw.pop("anaglyph")
    if "rgbd" in kw:
        kw["rgbd_out"] = kw.pop("rgbd")
    return kw


def _args(**kw):
    base = dict(ipd_offset=0, rgbd=False, half_rgbd=False, pad=None, pad_mode="tblr", vr180=False, half_sbs=False, half_tb=False,
                anaglyph=None, tb=False, cross_eyed=False, max_output_height=None, max_output_width=None,
                keep_aspect_ratio=False)
    base.update(kw)
    return argparse.Namespace(**base)


@pytest.mark.parametrize("name", sorted(FORMAT_CASES))
def test_oracle_matches_reference_fixture(g, name):
    out = OU.postprocess_image(g["left"].clone(), g["right"].clone(), **_oracle_kwargs(FORMAT_CASES[name]))
    assert out.shape == g[name].shape
    assert (out - g[name]).abs().max().item() < 2e-6, name


def test_oracle_rgbd(g):
    depth = OU.MAPPERS["pow2"](g["depth"])
    for name, kw in (("rgbd", dict(rgbd_out=True)), ("half_rgbd", dict(half_rgbd=True, ipd_offset=3.0))):
        le, re = OU.rgbd(g["left"], depth)
        out = OU.postprocess_image(le, re, **kw)
        assert out.shape == g[name].shape and (out - g[name]).abs().max().item() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FORMAT_CASES))
def test_hip_postprocess_image(hiplib, g, name):
    from nunif_amd.iw3.utils import postprocess_image
    left, right = g["left"].to("cuda:0"), g["right"].to("cuda:0")
    out = postprocess_image(left, right, _args(**FORMAT_CASES[name]))
    assert out.shape == g[name].shape, (out.shape, g[name].shape)
    p = psnr(out.cpu(), g[name])
    assert p >= 50.0, (name, p)
    if name.startswith("pad_") or name in ("ana_color", "sbs_ipd", "sbs_ipd_neg"):
        assert torch.equal(out.cpu(), g[name]), name            # pure data movement: bit-exact


@pytest.mark.gpu
def test_hip_rgbd_and_errors(hiplib, g):
    from nunif_amd.iw3.anaglyph import apply_anaglyph_redcyan
    from nunif_amd.iw3.utils import apply_rgbd, postprocess_image
    left, depth = g["left"].to("cuda:0"), g["depth"].to("cuda:0")
    for name, kw in (("rgbd", dict(rgbd=True)), ("half_rgbd", dict(half_rgbd=True, ipd_offset=3.0))):
        le, re = apply_rgbd(left, depth, mapper="pow2")
        assert re.shape == le.shape
        assert psnr(postprocess_image(le, re, _args(**kw)).cpu(), g[name]) >= 50.0
    with pytest.raises(ValueError):
        apply_anaglyph_redcyan(left, left, "nope")
    with pytest.raises(RuntimeError):
        apply_anaglyph_redcyan(g["left"], g["right"], "color")      # CPU tensors: no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(tb=True), dict(cross_eyed=True), dict(max_output_width=4096),
                                dict(half_sbs=True), dict(pad=0.1), dict(anaglyph="dubois"), dict(ipd_offset=2),
                                dict(max_output_width=100, keep_aspect_ratio=True)])
@pytest.mark.parametrize("use_16bit", [False, True])
def test_fused_leave_path_writes_the_same_bytes_as_compose_then_quantise(kw, use_16bit):
    """``postprocess_to_frame`` (the scheduler's default way out: compose + quantise in one kernel where the format allows) against
    ``to_frame_tensor(postprocess_image(...))`` on the same eyes: identical frames for plain and non-plain formats alike."""
    from nunif_amd.iw3 import utils as U
    gen = torch.Generator().manual_seed(77)
    left = (torch.rand(3, 54, 96, generator=gen) * 1.2 - 0.1).cuda()       # values outside [0, 1] exercise the clamp
    right = (torch.rand(3, 54, 96, generator=gen) * 1.2 - 0.1).cuda()
    args = _args(**kw)
    fused = U.postprocess_to_frame(left, right, args, use_16bit=use_16bit)
    two_step = U.to_frame_tensor(U.postprocess_image(left, right, args), use_16bit=use_16bit)
    assert fused.dtype == two_step.dtype and fused.shape == two_step.shape
    assert torch.equal(fused, two_step)


@pytest.mark.parametrize("kw,fused", [
    (dict(), True), (dict(tb=True), True), (dict(cross_eyed=True), True),
    (dict(max_output_width=4096, max_output_height=4096), True),       # caps that do not bite
    (dict(max_output_width=100), False), (dict(max_output_height=40), False),
    (dict(half_sbs=True), False), (dict(half_tb=True), False), (dict(half_rgbd=True), False), (dict(vr180=True), False),
    (dict(pad=0.1), False), (dict(pad_mode="16:9"), False), (dict(anaglyph="dubois"), False), (dict(ipd_offset=3), False),
])
def test_fused_leave_path_is_taken_exactly_for_the_plain_formats(kw, fused, monkeypatch):
    """Host logic only (no GPU): ``postprocess_to_frame`` may skip ``postprocess_image`` only where that function would do nothing
    but compose — every padding, projection, half-size, anaglyph or biting output cap must go the two-step way."""
    from nunif_amd.iw3 import utils as U
    calls = []
    monkeypatch.setattr(U._ops, "stereo_to_frame", lambda le, re, layout, bits: calls.append(("fused", layout, bits)) or "F")
    monkeypatch.setattr(U, "postprocess_image", lambda le, re, args: calls.append(("post",)) or "P")
    monkeypatch.setattr(U, "to_frame_tensor", lambda x, use_16bit=False: calls.append(("frame", x, use_16bit)) or "T")
    left, right = torch.zeros(3, 54, 96), torch.zeros(3, 54, 96)
    out = U.postprocess_to_frame(left, right, _args(**kw), use_16bit=True)
    if fused:
        layout = "tb" if kw.get("tb") else ("cross_eyed" if kw.get("cross_eyed") else "sbs")
        assert out == "F" and calls == [("fused", layout, 16)]
    else:
        assert out == "T" and calls == [("post",), ("frame", "P", True)]
