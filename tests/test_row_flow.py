"""sbs.row_flow_v3 + the NN backward warp: oracle vs the reference fixture (CPU), HIP engine vs fixture / oracle (GPU)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr, sd_checksum, synth_image
from oracle import row_flow_v3 as ORF
from oracle.forward_warp import synth_depth


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "row_flow.npz")).items()}


def test_oracle_matches_reference_fixture(g):
    sd = ORF.random_state_dict(301)
    assert sd_checksum(sd) == pytest.approx(float(g["sdsum"]), rel=1e-12)
    depth, c = g["depth"], g["c"]
    delta = ORF.delta_forward(sd, ORF.make_input(depth, 2.0, 0.5, 104))
    assert delta.shape == g["delta"].shape == (2, 1, 58, 104) and (delta - g["delta"]).abs().max().item() < 2e-5
    assert g["delta"].std().item() > 0.2                      # the fixture's flow really moves pixels
    left, right = ORF.apply_divergence_nn_LR(sd, c, depth, 2.0, 0.5)
    assert (left - g["left"]).abs().max().item() < 2e-4 and (right - g["right"]).abs().max().item() < 2e-4
    assert (left - c).abs().mean().item() > 1e-3
    _, ro = ORF.apply_divergence_nn_LR(sd, c[:1], depth[:1], 2.0, 0.5, synthetic_view="right")
    assert (ro - g["right_only"]).abs().max().item() < 2e-4
    ls, rs = ORF.apply_divergence_nn_LR(sd, c[:1, :, :58, :104].contiguous(), depth[:1], 2.0, 0.5)
    assert (ls - g["left_same"]).abs().max().item() < 2e-4 and (rs - g["right_same"]).abs().max().item() < 2e-4


def test_score_bias_buffers_match_reference_layout():
    index, delta = ORF.window_score_bias_input((4, 4))
    assert index.shape == (256,) and delta.shape == (49, 2) and index[:4].tolist() == [24, 23, 22, 21]
    index, delta = ORF.window_score_bias_input((3, 3))
    assert index.shape == (81,) and delta.shape == (25, 2) and float(delta.abs().max()) == 1.0


@pytest.mark.gpu
def test_hip_delta_and_warp(hiplib, g):
    from nunif_amd.iw3.models.row_flow_v3 import RowFlowV3
    from nunif_amd.iw3.backward_warp import apply_divergence_nn_LR, make_input_tensor
    sd = ORF.random_state_dict(301)
    m = RowFlowV3().eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    m.delta_output = True
    assert (m.name, m.i2i_scale, m.i2i_offset, m.i2i_blend_size) == ("sbs.row_flow_v3", 1, 32, 4)
    depth, c = g["depth"].to("cuda:0"), g["c"].to("cuda:0")
    x = torch.stack([make_input_tensor(None, depth[i], 2.0, 0.5, 104) for i in range(2)])
    d2 = m(x)
    assert d2.shape == (2, 2, 58, 104) and float(d2[:, 1].abs().max()) == 0.0
    err = (d2[:, :1].cpu() - g["delta"]).abs()
    # delta is in depth pixels; fp16 activations: a few 1e-3 px
    assert err.max().item() < 2e-2 and err.mean().item() < 2e-3, (err.max().item(), err.mean().item())
    left, right = apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=None, synthetic_view="both")
    assert psnr(left.cpu(), g["left"]) >= 50.0 and psnr(right.cpu(), g["right"]) >= 50.0, \
        (psnr(left.cpu(), g["left"]), psnr(right.cpu(), g["right"]))
    lo, ro = apply_divergence_nn_LR(m, c[:1], depth[:1], 2.0, 0.5, steps=1, synthetic_view="right")
    assert torch.equal(lo, c[:1]) and psnr(ro.cpu(), g["right_only"]) >= 50.0
    lb, rb = apply_divergence_nn_LR(m, c[:1], depth[:1], 2.5, 0.4, steps=1, preserve_screen_border=True)
    assert psnr(lb.cpu(), g["left_border"]) >= 50.0 and psnr(rb.cpu(), g["right_border"]) >= 50.0
    ls, rs = apply_divergence_nn_LR(m, c[:1, :, :58, :104].contiguous(), depth[:1], 2.0, 0.5, steps=1)
    assert psnr(ls.cpu(), g["left_same"]) >= 50.0 and psnr(rs.cpu(), g["right_same"]) >= 50.0
    # deterministic, batch-independent
    l1, _ = apply_divergence_nn_LR(m, c[1:], depth[1:], 2.0, 0.5, steps=1)
    assert torch.equal(l1, left[1:])


@pytest.mark.gpu
def test_hip_row_flow_1080p_vs_oracle(hiplib):
    """BASELINE config 4 geometry: 1080p frame, depth 392x686 (DepthAnything's output size), default method."""
    from nunif_amd.iw3.models.row_flow_v3 import RowFlowV3
    from nunif_amd.iw3.utils import apply_divergence
    sd = ORF.random_state_dict(302)
    m = RowFlowV3().eval()
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    m.delta_output = True
    depth = synth_depth(5, 1, 392, 686, "smooth_edges")
    c = synth_image(73, 3, 1080, 1920)[None]
    args = SimpleNamespace(mapper="none", convergence=0.5, divergence=2.0, method="row_flow_v3", synthetic_view="both",
                           warp_steps=None, stereo_width=None, preserve_screen_border=False, disable_amp=False)
    left, right = apply_divergence(depth.to("cuda:0"), c.to("cuda:0"), args, side_model=m)
    lo, ro = ORF.apply_divergence_nn_LR(sd, c, depth, 2.0, 0.5)
    assert left.shape == (1, 3, 1080, 1920)
    assert psnr(left.cpu(), lo) >= 50.0 and psnr(right.cpu(), ro) >= 50.0, (psnr(left.cpu(), lo), psnr(right.cpu(), ro))
    cpu_model = RowFlowV3().eval()
    cpu_model.delta_output = True
    with pytest.raises(RuntimeError):
        cpu_model(torch.rand(1, 3, 64, 64))     # a CPU-resident model has no engine: no fallback


# ---- symmetric use (row_flow_v3_sym: apply_divergence_nn_symmetric) -----------------------------------------------------
@pytest.fixture(scope="module")
def gs():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "row_flow_sym.npz")).items()}


def test_oracle_symmetric_matches_reference_fixture(gs):
    sd = ORF.random_state_dict(311)
    left, right = ORF.apply_divergence_nn_symmetric(sd, gs["c"], gs["depth"], 2.0, 0.5, "both")
    assert (left - gs["left"]).abs().max().item() < 2e-4 and (right - gs["right"]).abs().max().item() < 2e-4
    _, r = ORF.apply_divergence_nn_symmetric(sd, gs["c"], gs["depth"], 2.0, 0.5, "right")
    l, c = ORF.apply_divergence_nn_symmetric(sd, gs["c"], gs["depth"], 2.0, 0.5, "left")
    assert (r - gs["right_only"]).abs().max().item() < 2e-4 and (l - gs["left_only"]).abs().max().item() < 2e-4
    assert c is gs["c"]


@pytest.mark.gpu
def test_hip_row_flow_symmetric(hiplib, gs):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    from nunif_amd.iw3.backward_warp import apply_divergence_nn_LR
    m = create_model("sbs.row_flow_v3").eval()
    m.load_state_dict(ORF.random_state_dict(311), strict=True)
    m = m.to("cuda:0")
    m.delta_output, m.symmetric = True, True
    c, depth = gs["c"].to("cuda:0"), gs["depth"].to("cuda:0")
    left, right = apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=1, synthetic_view="both")
    assert psnr(left.cpu(), gs["left"]) >= 50.0 and psnr(right.cpu(), gs["right"]) >= 50.0
    le, r = apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=1, synthetic_view="right")
    assert le is c and psnr(r.cpu(), gs["right_only"]) >= 50.0
    l, ri = apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=1, synthetic_view="left")
    assert ri is c and psnr(l.cpu(), gs["left_only"]) >= 50.0


# ---- warp_steps > 1 (iw3/backward_warp.py:190-231) --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gst():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "row_flow_steps.npz")).items()}


def test_oracle_warp_steps_match_reference_fixture(gst):
    sd = ORF.random_state_dict(301)
    assert sd_checksum(sd) == pytest.approx(float(gst["sdsum"]), rel=1e-12)
    depth, c = gst["depth"], gst["c"]
    for steps, div in ((2, 6.0), (3, 9.0)):
        left, right = ORF.apply_divergence_nn_LR(sd, c, depth, div, 0.5, steps=steps)
        assert (left - gst[f"left_s{steps}"]).abs().max().item() < 5e-4
        assert (right - gst[f"right_s{steps}"]).abs().max().item() < 5e-4
        one, _ = ORF.apply_divergence_nn_LR(sd, c, depth, div, 0.5, steps=1)
        assert (one - left).abs().mean().item() > 1e-3            # the stepped warp really differs from the single one


def test_calc_auto_warp_steps_matches_reference_table(gst):
    from nunif_amd.iw3.utils import calc_auto_warp_steps
    tab = gst["auto_steps"].tolist()
    for d, want in tab[:8]:
        assert (calc_auto_warp_steps("row_flow_v3", d, "both") or 0) == int(want), d
    for d, want in tab[8:]:
        assert (calc_auto_warp_steps("row_flow_v3", d, "right") or 0) == int(want), d
    assert calc_auto_warp_steps("row_flow", 6.0, "both") == 2 and calc_auto_warp_steps("mlbw_l2", 20.0, "both") is None
    assert calc_auto_warp_steps("row_flow_v2", 2.6, "both") == 2


@pytest.mark.gpu
def test_hip_warp_steps(hiplib, gst):
    """steps 2 and 3 on the engine against the reference's own outputs (>= 50 dB), the mirror of the right eye folded into
    the kernels for every step; the single-eye + screen-border case against the oracle."""
    from nunif_amd.iw3.models.row_flow_v3 import RowFlowV3
    from nunif_amd.iw3.backward_warp import apply_divergence_nn_LR
    sd = ORF.random_state_dict(301)
    m = RowFlowV3().eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    m.delta_output = True
    depth, c = gst["depth"].to("cuda:0"), gst["c"].to("cuda:0")
    for steps, div in ((2, 6.0), (3, 9.0)):
        left, right = apply_divergence_nn_LR(m, c, depth, div, 0.5, steps=steps, synthetic_view="both")
        pl, pr = psnr(left.cpu(), gst[f"left_s{steps}"]), psnr(right.cpu(), gst[f"right_s{steps}"])
        assert pl >= 50.0 and pr >= 50.0, (steps, pl, pr)
    le, ro = apply_divergence_nn_LR(m, c[:1], depth[:1], 3.0, 0.4, steps=2, synthetic_view="right", preserve_screen_border=True)
    assert torch.equal(le, c[:1]) and psnr(ro.cpu(), gst["right_only_s2"]) >= 50.0
