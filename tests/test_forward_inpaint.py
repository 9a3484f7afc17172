"""``--method forward_inpaint`` (iw3/forward_inpaint.py): oracle vs the reference fixture on CPU, HIP engine vs the fixture on
the GPU.  ``tests/golden/forward_inpaint.npz`` comes from the reference's own ForwardInpaintImage / ForwardInpaintVideo
(``make_golden.py::gen_forward_inpaint``) around seeded inpaint nets."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import light_inpaint as OL


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "forward_inpaint.npz")).items()}


def _batches(g):
    f, d = g["v_frames"], g["v_depth"]
    return [(f[i:i + 3], d[i:i + 3]) for i in range(0, f.shape[0], 3)]


def test_oracle_image_flow(g):
    sd = OL.random_state_dict(701)
    c, depth = g["c"], g["depth"]
    le, ri = OL.forward_inpaint_image(sd, c, depth, 2.5, 0.5, "both", 1, 2)
    assert (le - g["fi_left"].float()).abs().max().item() < 2e-3 and (ri - g["fi_right"].float()).abs().max().item() < 2e-3
    le, ri = OL.forward_inpaint_image(sd, c[:1], depth[:1], 2.0, 0.3, "right")
    assert torch.equal(le, c[:1]) and (ri - g["fi_right_only"].float()).abs().max().item() < 2e-3
    le, ri = OL.forward_inpaint_image(sd, c[1:], depth[1:], 2.0, 0.5, "left", max_width=150)
    assert le.shape == g["fi_left_mw"].shape == (1, 3, 84, 150)
    assert (le - g["fi_left_mw"].float()).abs().max().item() < 2e-3 and (ri - g["fi_right_mw"].float()).abs().max().item() < 2e-3


def test_oracle_video_queue_flow(g):
    sdv = OL.video_random_state_dict(801)
    res = OL.forward_inpaint_video(sdv, _batches(g)[:3], 2.0, 0.5, 1, 1)          # first output + flush (CPU time)
    assert [0 if r is None else r[0].shape[0] for r in res] == [0, 0, 6, 3]
    left = torch.cat([r[0] for r in res if r is not None])
    assert (left[:6] - g["v_left"][:6].float()).abs().max().item() < 2e-3      # fixture stored as fp16


def _models(video=False):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    mi = create_model("inpaint.light_inpaint_v1").eval()
    mi.load_state_dict(OL.random_state_dict(701), strict=True)
    mv = None
    if video:
        mv = create_model("inpaint.light_video_inpaint_v1").eval()
        mv.load_state_dict(OL.video_random_state_dict(801), strict=True)
        mv = mv.to("cuda:0")
    return mi.to("cuda:0"), mv


def _close(got, ref, key):
    # the warp and the masks are bit-exact; the inpaint net computes in fp16: same criterion as the MLBW inpaint tests
    assert got.shape == ref.shape, key
    bad = ((got.cpu() - ref.float()).abs() > 2e-2).float().mean().item()
    assert bad < 0.03, (key, bad)


@pytest.mark.gpu
def test_hip_forward_inpaint_image(hiplib, g):
    from nunif_amd.iw3.forward_inpaint import ForwardInpaint
    side = ForwardInpaint(_models()[0])
    side.set_mode("image")
    c, depth = g["c"].to("cuda:0"), g["depth"].to("cuda:0")
    le, ri = side.infer(c, depth, divergence=2.5, convergence=0.5, synthetic_view="both", inner_dilation=1, outer_dilation=2)
    _close(le, g["fi_left"], "fi_left"); _close(ri, g["fi_right"], "fi_right")
    le, ri = side.infer(c[:1], depth[:1], divergence=2.0, convergence=0.3, synthetic_view="right")
    assert torch.equal(le, c[:1])
    _close(ri, g["fi_right_only"], "fi_right_only")
    le, ri = side.infer(c[1:], depth[1:], divergence=2.0, convergence=0.5, synthetic_view="left", max_width=150)
    _close(le, g["fi_left_mw"], "fi_left_mw"); _close(ri, g["fi_right_mw"], "fi_right_mw")
    assert side.flush() == (None, None)
    with pytest.raises(NotImplementedError):
        side.set_mode("video")


@pytest.mark.gpu
def test_hip_forward_inpaint_video_queue_and_method_dispatch(hiplib, g):
    from nunif_amd.iw3.forward_inpaint import ForwardInpaint
    from nunif_amd.iw3.utils import apply_divergence
    side = ForwardInpaint(*_models(video=True))
    side.set_mode("video")
    sizes, lefts, rights = [], [], []
    for f, d in _batches(g):
        le, ri = side.infer(f.to("cuda:0"), d.to("cuda:0"), divergence=2.0, convergence=0.5, synthetic_view="both",
                            inner_dilation=1, outer_dilation=1)
        sizes.append(0 if le is None else le.shape[0])
        if le is not None:
            lefts.append(le.clone()); rights.append(ri.clone())
    le, ri = side.flush()
    sizes.append(0 if le is None else le.shape[0])
    lefts.append(le); rights.append(ri)
    assert sizes == [int(v) for v in g["v_sizes"]] == [0, 0, 6, 0, 6, 3]
    _close(torch.cat(lefts), g["v_left"], "v_left"); _close(torch.cat(rights), g["v_right"], "v_right")
    assert side.flush() == (None, None)
    side.reset()
    # iw3.utils.apply_divergence's forward_inpaint / mlbw_l2_inpaint branch (:333-365): one side_model.infer per frame, a
    # flush where reset_pts says so, results concatenated (None while the queue fills)
    args = SimpleNamespace(mapper="none", convergence=0.5, divergence=2.0, method="forward_inpaint", synthetic_view="both",
                           preserve_screen_border=False, mask_inner_dilation=1, mask_outer_dilation=1, inpaint_max_width=None,
                           disable_amp=False)
    f, d = g["v_frames"].to("cuda:0"), g["v_depth"].to("cuda:0")
    le, ri = apply_divergence(d[:3], f[:3], args, side_model=side)
    assert le is None and ri is None                          # 3 + 3 frames in the queue
    le, ri = apply_divergence(d[3:9], f[3:9], args, side_model=side, reset_pts=[False] * 5 + [True])
    assert le.shape[0] == 9 and ri.shape[0] == 9              # 6 when the queue fills + the flush at the cut: all 9 frames
    _close(le[:6], g["v_left"][:6], "dispatch_left")
    side.set_mode("image")
    le, ri = apply_divergence(g["depth"].to("cuda:0"), g["c"].to("cuda:0"),
                              SimpleNamespace(**{**vars(args), "divergence": 2.5, "mask_outer_dilation": 2}), side_model=side)
    _close(le, g["fi_left"], "dispatch_image_left")
