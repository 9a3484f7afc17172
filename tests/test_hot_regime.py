"""Trained-regime parity stress.

Every other fixture uses weights conditioned into a BENIGN regime (``nunif_amd/synthetic.py``: damped residual branches, tiny
relative-position tables) because no trained checkpoint exists offline.  ``regime="hot"`` puts the same nets where trained ones
live — relative-position tables N(0, 1.5), undamped residual streams (|stream| 50-160, logits tens of units), outlier channels,
DINOv2-style massive activations and a common offset in front of LayerNorm, inputs with saturated flats — and, because NO fp16
engine is within 50 dB of fp32 there, holds the HIP engine to the reference's own GPU arithmetic instead: the fp32 oracle re-run
with every op result rounded to fp16 (``oracle/fp16_emulation.py`` = ``torch.autocast(cuda, float16)``, ``nunif/device.py:58-71``,
emulated, parameters rounded to fp16 like autocast does).  Criterion:  PSNR(hip, fp32 reference) >= PSNR(emulated fp16
reference, fp32 reference) - margin,  and every value finite.

Margins.  cunet / depth ViT-S: 1 dB (measured: cunet +0.4 dB, i.e. BETTER than the emulation; ViT-S rel. rms 3.3e-3 against the
emulation's 4.3e-3).  swin_unet: 3 dB = at most twice the emulated reference's noise power.  Measured on the GPU (round 4):
    case        emulated fp16 ref   HIP (packed-fp16 GELU)   HIP (fp32-polynomial GELU build)
    2x          45.42 dB            44.58                    44.69
    2x_chaos    26.82               24.44                    24.41      (37 % of the picture clamped: chaotic)
    4x          44.67               44.22                    43.52
    1x          42.68               39.99                    41.28
Two builds that differ only in HOW the same GELU is evaluated move by +0.7 / -1.3 dB against each other in this regime, so a
1 dB margin (VERDICT's suggestion) is inside the regime's own noise; the mean gap to the emulation is ~1.5 dB in both builds.
(The first version of the emulation left the parameters in fp32 and read 3 dB better — that 3 dB is what fp16 WEIGHTS cost any
fp16 engine, the reference's autocast included.)

Round 5: the margin is no longer read off one sample.  ``tools/hot_regime_stats.py`` runs every case's weights
on 8 inputs of 64 x 64 and one 256 x 256 tile and reports gap = PSNR(HIP) - PSNR(emulation), both against the fp32 oracle
(``profiles/r05a_hot_packed.json``, ``r05a_hot_gelu32.json``, ``r05b_hot.json``):
    case        mean gap +- std (worst), packed-fp16 GELU     fp32-polynomial GELU build      256 x 256 tile (packed / fp32)
    2x          -0.77 +- 1.21 (-2.15)                         -0.82 +- 1.37 (-2.22)           -0.99 / -1.80
    2x_chaos    -2.81 +- 0.49 (-3.52)                         -2.76 +- 0.52 (-3.74)           -1.18 / -1.40
    4x          -0.86 +- 0.81 (-1.99)                         -0.56 +- 0.90 (-2.03)           -0.54 / -0.81
    1x          -1.81 +- 1.03 (-3.65)                         -0.55 +- 0.98 (-2.26)           -1.64 / +0.41
A single input scatters by +-1 dB (one sigma) around its case's mean, so a per-sample margin below ~2.5 sigma + |mean| fails at
random: the single-sample tests keep 3 dB and the CRITERION is the mean over 8 inputs, >= -1.5 dB
(``test_hip_swin_unet_hot_regime_gap_statistics``).  The packed GELU costs the 1x net 1.3 dB (6 sigma of the mean) and nothing
measurable elsewhere: the 1x net now runs the fp32-polynomial form (``swin_gelu.h`` ``gelu8t<G32>``; -0.51 +- 0.95 in ``r05b_hot.json``).
``2x_chaos`` is the one case outside 1.5 dB: 37 % of that picture is clamped and the emulation ITSELF is at 26.8 dB — two fp16
evaluation orders of the same net end up 2-3 dB apart there (the emulation rounds after every op; the engine keeps fp32 inside fused
blocks and rounds at kernel boundaries), its bound is -3.5 dB on the mean and it is reported, not hidden.

``tests/golden/hot_regime.npz`` (``make_golden.py hot``) holds, per case, the REFERENCE's fp32 output and the emulated one.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, hot_image, psnr, sd_checksum
from oracle import cunet as OC
from oracle import depth_anything_v2 as OD
from oracle import swin_unet as O
from oracle.fp16_emulation import fp16_autocast_emulation

# (tag, scale factor, seed) — tests/golden/make_golden.py::HOT_SWIN
HOT_SWIN = (("2x", 2, 432), ("2x_chaos", 2, 422), ("4x", 4, 404), ("1x", 1, 411))
MARGIN_DB = 1.0            # cunet, depth
SWIN_MARGIN_DB = 3.0       # see the module docstring


@pytest.fixture(scope="module")
def hot():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "hot_regime.npz")).items()}


def _mse(a, b):
    return torch.mean((a.double() - b.double()) ** 2).item()


def test_hot_weights_are_hot_and_the_oracle_still_is_the_reference(hot):
    """CPU: the hot fixtures are what they claim (the emulated-fp16 reference is 10-30 dB below the benign regime's 59.8), the
    fp32 oracle still equals the reference's output on them, and the inputs carry exact 0 / 1 flats."""
    x = torch.stack([hot_image(21, 64, 64), hot_image(22, 64, 64)])
    assert torch.equal(x, hot["swin_x"]) and float((x == 1).float().mean()) > 0.1 and float((x == 0).float().mean()) > 0.05
    for tag, sf, seed in HOT_SWIN:
        sd = O.random_state_dict(seed, sf, regime="hot")
        assert sd_checksum(sd) == pytest.approx(float(hot[f"swin_{tag}_sdsum"]), rel=1e-12)
        assert sd["unet.swin1.block.0.attn.relative_position_bias_table"].std().item() > 1.0
        ref, emu = hot[f"swin_{tag}_ref"], hot[f"swin_{tag}_emu"]
        assert torch.isfinite(emu).all() and 20.0 < psnr(emu, ref) < 50.0, (tag, psnr(emu, ref))
        if tag == "2x":
            taps = {}
            y = torch.clamp(O.unet_forward(sd, x, sf, taps=taps), 0, 1)
            assert _mse(y, ref) < 1e-7
            assert max(v.abs().max().item() for k, v in taps.items() if k.endswith(".out")) > 30.0     # an undamped stream
    sd = OC.random_state_dict(601, up=False, regime="hot")
    assert sd_checksum(sd) == pytest.approx(float(hot["cunet_sdsum"]), rel=1e-12)
    assert (OC.model_forward(sd, hot["cunet_x"]) - hot["cunet_ref"]).abs().max().item() < 1e-4
    sd = OD.random_state_dict(301, grid=37, regime="hot")
    assert sd_checksum(sd) == pytest.approx(float(hot["depth_sdsum"]), rel=1e-12)
    feats, _, _ = OD.encoder_features(sd, hot["depth_x"])
    rel = ((hot["depth_emu"] - hot["depth_ref"]).pow(2).mean().sqrt() / hot["depth_ref"].std()).item()
    assert 5e-4 < rel < 2e-2, rel


def test_emulation_rounds_every_result_and_nothing_else():
    a = torch.tensor([1.0 + 2.0 ** -12, 70000.0, 3.0])
    with fp16_autocast_emulation():
        b = a + 0.0
        c = torch.arange(5)
        w = torch.nn.functional.linear(torch.ones(1, 2048), torch.full((1, 2048), 1.0 + 2.0 ** -10))
    assert b[0].item() == 1.0 and torch.isinf(b[1]) and b[2].item() == 3.0 and c.dtype == torch.int64
    # operands are rounded by the ops that produced them, the accumulation itself is fp32: 2048 x (1 + 2^-10) = 2050 exactly
    # (an fp16 accumulator would stall at 2048, where its spacing is 2)
    assert w.item() == 2050.0
    assert (a + 0.0)[0].item() != 1.0                       # and outside the context nothing is touched


def _criterion(tag, y, ref, emu, margin=MARGIN_DB):
    assert y.shape == ref.shape and torch.isfinite(y).all(), tag
    p_hip, p_emu = psnr(y, ref), psnr(emu, ref)
    print(f"\nhot {tag}: HIP vs fp32 reference {p_hip:.2f} dB, emulated fp16 reference vs fp32 {p_emu:.2f} dB")
    assert p_hip >= p_emu - margin, (tag, p_hip, p_emu)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,sf,seed", HOT_SWIN)
def test_hip_swin_unet_in_the_hot_regime(hiplib, hot, capsys, tag, sf, seed):
    from nunif_amd.waifu2x.models import swin_unet as M
    sd = O.random_state_dict(seed, sf, regime="hot")
    m = {1: M.SwinUNet, 2: M.SwinUNet2x, 4: M.SwinUNet4x}[sf]().eval()
    m.load_state_dict(sd, strict=True)
    y = m.to("cuda:0")(hot["swin_x"].to("cuda:0")).cpu()
    with capsys.disabled():
        _criterion(f"swin_{tag}", y, hot[f"swin_{tag}_ref"], hot[f"swin_{tag}_emu"], SWIN_MARGIN_DB)


MEAN_GAP_MIN_DB = {"2x": -1.5, "4x": -1.5, "1x": -1.5, "2x_chaos": -3.5}      # module docstring
WORST_GAP_MIN_DB = -4.5                                                       # mean - 3 sigma of the noisiest case


@pytest.mark.gpu
@pytest.mark.parametrize("tag,sf,seed", HOT_SWIN)
def test_hip_swin_unet_hot_regime_gap_statistics(hiplib, capsys, tag, sf, seed):
    """Mean over 8 inputs (and one 256 x 256 tile) of PSNR(HIP) - PSNR(emulated fp16 reference), both against the fp32 oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import hot_regime_stats as H
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    st = H.summarize(H.case_gaps(tag, sf, seed, n_inputs=8, big=True))
    with capsys.disabled():
        print(f"\nhot swin_{tag}: gap to the emulated fp16 reference over {st['n']} inputs: mean {st['mean_gap_db']:+.2f} dB, "
              f"std {st['std_gap_db']:.2f}, worst {st['worst_gap_db']:+.2f}; 256 x 256 tile {st['tile256']['gap']:+.2f}")
    assert st["mean_gap_db"] >= MEAN_GAP_MIN_DB[tag], st
    assert st["worst_gap_db"] >= WORST_GAP_MIN_DB and st["tile256"]["gap"] >= WORST_GAP_MIN_DB, st


@pytest.mark.gpu
def test_hip_cunet_in_the_hot_regime(hiplib, hot, capsys):
    from nunif_amd.waifu2x.models.cunet import CUNet
    m = CUNet().eval()
    m.load_state_dict(OC.random_state_dict(601, up=False, regime="hot"), strict=True)
    y = m.to("cuda:0")(hot["cunet_x"].to("cuda:0")).cpu()
    with capsys.disabled():
        _criterion("cunet", y, hot["cunet_ref"], hot["cunet_emu"])


@pytest.mark.gpu
def test_hip_depth_vits_with_massive_activations(hiplib, hot, capsys):
    """DINOv2-style outlier channels (+-45 in 2 of 384) and a +12 common offset in front of every LayerNorm: the LN-folded
    Linears take their variance as E[x^2] - mean^2 from fp32 partial sums of the fp16-stored rows."""
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    net = HipDepthAnythingV2(OD.random_state_dict(301, grid=37, regime="hot"), "cuda:0")
    y = net(hot["depth_x"].to("cuda:0")).cpu()
    ref, emu = hot["depth_ref"], hot["depth_emu"]
    assert y.shape == ref.shape and torch.isfinite(y).all()
    rel_hip = ((y - ref).pow(2).mean().sqrt() / ref.std()).item()
    rel_emu = ((emu - ref).pow(2).mean().sqrt() / ref.std()).item()
    with capsys.disabled():
        print(f"\nhot depth ViT-S: HIP rel. rms {rel_hip:.2e}, emulated fp16 reference {rel_emu:.2e}")
    assert rel_hip <= rel_emu * 10 ** (MARGIN_DB / 20) + 1e-4, (rel_hip, rel_emu)


# ---- the iw3 side nets in a hot regime (only swin / cunet / ViT-S had one) -------------------------------------------
# Weights: ``regime="hot"`` of the same generators (nunif_amd/synthetic.py: undamped residual branches, window logits of tens of
# units, spread LayerNorm gains).  The fp32 oracle and its fp16-autocast emulation run on the host at test time (small maps);
# the oracle is pinned to the reference on the benign weights by the committed fixtures (test_row_flow / test_mlbw /
# test_light_inpaint) and, on THESE weights, live when the reference is mounted (``test_hot_iw3_oracles_are_the_reference``).
# Measured on the CPU (emulated fp16 reference vs fp32): row_flow_v3 delta rms error 3.2e-4 (benign) -> 3.2e-3 (hot), mlbw_l2
# 7.5e-4 -> 7.8e-3, light_inpaint_v1 59.8 dB -> 40.4 dB.
IW3_MARGIN_DB = 1.5


def _rms(a, b):
    return (a.double() - b.double()).pow(2).mean().sqrt().item()


def _hot_iw3_inputs():
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "row_flow.npz")).items()}
    return g["depth"]


@pytest.mark.gpu
def test_hip_row_flow_v3_in_the_hot_regime(hiplib, capsys):
    from oracle import row_flow_v3 as ORF
    from oracle.fp16_emulation import half_weights
    from nunif_amd.iw3.models.row_flow_v3 import RowFlowV3
    from nunif_amd.iw3.backward_warp import make_input_tensor
    sd = ORF.random_state_dict(301, regime="hot")
    depth = _hot_iw3_inputs()
    x = ORF.make_input(depth, 2.0, 0.5, 104)
    ref = ORF.delta_forward(sd, x)
    with fp16_autocast_emulation():
        emu = ORF.delta_forward(half_weights(sd), x)
    m = RowFlowV3().eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    m.delta_output = True
    d = m(torch.stack([make_input_tensor(None, depth[i].to("cuda:0"), 2.0, 0.5, 104) for i in range(depth.shape[0])]))[:, :1].cpu()
    e_hip, e_emu = _rms(d, ref), _rms(emu, ref)
    with capsys.disabled():
        print(f"\nhot row_flow_v3: delta std {ref.std():.2f} px; rms error HIP {e_hip:.2e}, emulated fp16 reference {e_emu:.2e}")
    assert torch.isfinite(d).all() and ref.std().item() > 2.0
    assert e_hip <= e_emu * 10 ** (IW3_MARGIN_DB / 20) + 1e-4, (e_hip, e_emu)


@pytest.mark.gpu
def test_hip_mlbw_l2_in_the_hot_regime(hiplib, capsys):
    from oracle import mlbw as OM
    from oracle import row_flow_v3 as ORF
    from oracle.fp16_emulation import half_weights
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    from nunif_amd.iw3.backward_warp import make_input_tensor
    sd = OM.random_state_dict(402, 2, False, regime="hot")
    depth = _hot_iw3_inputs()
    x = ORF.make_input(depth[:1], 2.0, 0.5, 104)
    ref_d, ref_w = OM.delta_forward(sd, x, 2)
    with fp16_autocast_emulation():
        emu_d, emu_w = OM.delta_forward(half_weights(sd), x, 2)
    m = create_model("sbs.mlbw_l2").eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    m.delta_output = True
    d, w = m(torch.stack([make_input_tensor(None, depth[0].to("cuda:0"), 2.0, 0.5, 104)]))
    d, w = d.cpu(), w.cpu()
    with capsys.disabled():
        print(f"\nhot mlbw_l2: delta std {ref_d.std():.2f} px; delta rms error HIP {_rms(d, ref_d):.2e} / emulated {_rms(emu_d, ref_d):.2e}; "
              f"layer weight HIP {_rms(w, ref_w):.2e} / emulated {_rms(emu_w, ref_w):.2e}")
    assert torch.isfinite(d).all() and torch.isfinite(w).all() and ref_d.std().item() > 2.0
    assert _rms(d, ref_d) <= _rms(emu_d, ref_d) * 10 ** (IW3_MARGIN_DB / 20) + 1e-4
    assert _rms(w, ref_w) <= _rms(emu_w, ref_w) * 10 ** (IW3_MARGIN_DB / 20) + 1e-4


@pytest.mark.gpu
def test_hip_light_inpaint_v1_in_the_hot_regime(hiplib, capsys):
    from oracle import light_inpaint as OL
    from oracle.fp16_emulation import half_weights
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "light_inpaint.npz")).items()}
    sd = OL.random_state_dict(701, regime="hot")
    ref = OL.infer(sd, g["x"], g["mask"])
    with fp16_autocast_emulation():
        emu = OL.infer(half_weights(sd), g["x"], g["mask"])
    m = create_model("inpaint.light_inpaint_v1").eval()
    m.load_state_dict(sd, strict=True)
    y = m.to("cuda:0").infer(g["x"].to("cuda:0"), g["mask"].to("cuda:0")).cpu()
    with capsys.disabled():
        _criterion("light_inpaint_v1", y, ref, emu, IW3_MARGIN_DB)
    assert psnr(emu, ref) < 50.0                       # the regime is hot: fp16 storage alone is below the absolute bar


def test_hot_iw3_oracles_are_the_reference():
    """CPU, reference mounted: on the HOT weights the oracles of the three iw3 nets still equal the reference's own modules."""
    from oracle import refstub
    if not refstub.reference_available():
        pytest.skip("/root/reference is not mounted here")
    refstub.install()
    from oracle import light_inpaint as OL
    from oracle import mlbw as OM
    from oracle import row_flow_v3 as ORF
    import iw3.models  # noqa: F401  (registers sbs.* / inpaint.*)
    from nunif.models import create_model
    depth = _hot_iw3_inputs()
    x = ORF.make_input(depth, 2.0, 0.5, 104)
    with torch.inference_mode():
        sd = ORF.random_state_dict(301, regime="hot")
        m = create_model("sbs.row_flow_v3").eval()
        m.load_state_dict(sd, strict=True)
        m.delta_output = True
        assert (m(x)[:, :1] - ORF.delta_forward(sd, x)).abs().max().item() < 1e-3
        sd = OM.random_state_dict(402, 2, False, regime="hot")
        m = create_model("sbs.mlbw_l2").eval()
        m.load_state_dict(sd, strict=True)
        m.delta_output = True
        d, w = m(x[:1])
        od, ow = OM.delta_forward(sd, x[:1], 2)
        assert (d - od).abs().max().item() < 2e-3 and (w - ow).abs().max().item() < 1e-4
        g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "light_inpaint.npz")).items()}
        sd = OL.random_state_dict(701, regime="hot")
        m = create_model("inpaint.light_inpaint_v1").eval()
        m.load_state_dict(sd, strict=True)
        assert (m.infer(g["x"], g["mask"]) - OL.infer(sd, g["x"], g["mask"])).abs().max().item() < 1e-4
