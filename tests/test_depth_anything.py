"""Depth-Anything-V2 ViT-S backbone: HIP engine vs the architecture restatement (oracle/depth_anything_v2.py).

The real network lives in an external hub repository that is not available offline.  The restatement is
pinned against HuggingFace's independent ``DepthAnythingForDepthEstimation`` (tests/test_depth_anything_vs_hf.py), and
``test_hip_backbone_vs_huggingface_fixture`` compares the engine with HF-produced outputs directly."""
import os

import pytest
import torch

from conftest import psnr, synth_image
from oracle import depth_anything_v2 as ODA


def test_oracle_shapes_and_parameter_count():
    sd = ODA.random_state_dict(601)
    assert abs(sum(v.numel() for v in sd.values()) / 1e6 - 24.78) < 0.01          # Depth-Anything-V2-Small: 24.8 M
    x = torch.randn(2, 3, 56, 84)
    y = ODA.model_forward(sd, x)
    assert y.shape == (2, 56, 84) and float(y.min()) >= 0 and float(y.std()) > 0.1
    pe = ODA.interpolate_pos_embed(sd["pretrained.pos_embed"], 4, 6)
    assert pe.shape == (1, 25, 384) and torch.equal(pe[:, 0], sd["pretrained.pos_embed"][:, 0])


@pytest.mark.parametrize("encoder,mparams", [("vitb", 97.5), ("vitl", 335.3)])
def test_oracle_larger_encoders(encoder, mparams):
    """Published sizes: Depth-Anything-V2-Base 97.5 M, -Large 335.3 M parameters; V1 taps and the metric head run."""
    sd = ODA.random_state_dict(611, grid=4, encoder=encoder)
    n = sum(v.numel() for k, v in sd.items() if k != "pretrained.pos_embed") + 1370 * sd["pretrained.pos_embed"].shape[-1]
    assert abs(n / 1e6 - mparams) < 0.15, n / 1e6
    cfg = ODA.config_of(sd)
    assert cfg["taps"] == ((2, 5, 8, 11) if encoder == "vitb" else (4, 11, 17, 23)) and cfg["heads"] * 64 == cfg["embed"]
    x = torch.randn(1, 3, 28, 42)
    y = ODA.model_forward(sd, x)
    last4 = tuple(range(cfg["depth"] - 4, cfg["depth"]))
    y1 = ODA.model_forward(sd, x, taps=last4)
    ym = ODA.model_forward(sd, x, max_depth=20.0)
    assert y.shape == y1.shape == ym.shape == (1, 28, 42)
    assert float((y - y1).abs().max()) > 1e-3 and 0.0 < float(ym.min()) and float(ym.max()) < 20.0


def _norm(img):
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (img - mean) / std


@pytest.mark.gpu
def test_hip_backbone_vs_restatement(hiplib):
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    sd = ODA.random_state_dict(601)
    net = HipDepthAnythingV2(sd, "cuda:0")
    for shape in ((2, 3, 84, 126), (1, 3, 56, 56)):
        x = _norm(torch.stack([synth_image(90 + i, 3, shape[2], shape[3]) for i in range(shape[0])]))
        ref = ODA.model_forward(sd, x)
        y = net(x.to("cuda:0")).cpu()
        assert y.shape == ref.shape
        span = float(ref.max() - ref.min())
        p = psnr(y / span, ref / span)
        rel = ((y - ref).pow(2).mean().sqrt() / ref.std()).item()
        assert p >= 50.0 and rel < 1e-2, (shape, p, rel)
    assert torch.equal(net(x.to("cuda:0")).cpu(), y)                        # deterministic
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 50, 56))


@pytest.mark.gpu
def test_hip_backbone_vs_huggingface_fixture(hiplib):
    """The HIP backbone against ``tests/golden/depth_anything_hf.npz`` — outputs of ``transformers``'
    ``DepthAnythingForDepthEstimation`` (``oracle/hf_pin.py``, ``tests/golden/make_golden_hf.py``), an implementation our
    restatement had no part in: the 2 x 56 x 70 batch, the 392 x 686 map of a 1080p frame, the metric head and V1's taps."""
    import os
    import numpy as np
    from conftest import GOLDEN, sd_checksum
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    z = np.load(os.path.join(GOLDEN, "depth_anything_hf.npz"))
    sd = ODA.random_state_dict(601)
    assert sd_checksum(sd) == pytest.approx(float(z["sdsum"]), rel=1e-12)
    net = HipDepthAnythingV2(sd, "cuda:0")

    def check(y, ref, rel_max=1e-2):
        ref = torch.from_numpy(ref)
        assert y.shape == ref.shape
        span = float(ref.max() - ref.min())
        p = psnr(y / span, ref / span)
        rel = ((y - ref).pow(2).mean().sqrt() / ref.std()).item()
        assert float(ref.std()) > 1e-3 and p >= 50.0 and rel < rel_max, (p, rel)

    check(net(torch.from_numpy(z["x_small"]).to("cuda:0")).cpu(), z["y_small"])
    x = _norm(synth_image(95, 3, 392, 686)[None])
    check(net(x.to("cuda:0")).cpu(), z["y_392x686_seed95"])
    sd8 = ODA.random_state_dict(620, grid=8, encoder="vits")
    assert sd_checksum(sd8) == pytest.approx(float(z["sdsum8"]), rel=1e-12)
    x = _norm(torch.stack([synth_image(190 + i, 3, 70, 98) for i in range(2)])).to("cuda:0")
    check(HipDepthAnythingV2(sd8, "cuda:0", max_depth=80.0)(x).cpu(), z["y_metric80"], 1.5e-2)
    check(HipDepthAnythingV2(sd8, "cuda:0", taps=(8, 9, 10, 11))(x).cpu(), z["y_v1taps"], 1.5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("encoder,taps,max_depth", [("vitb", None, 0.0), ("vitl", None, 0.0), ("vitl", (20, 21, 22, 23), 0.0),
                                                    ("vitb", None, 20.0), ("vits", None, 80.0), ("vits", (8, 9, 10, 11), 0.0)])
def test_hip_backbone_variants_vs_restatement(hiplib, encoder, taps, max_depth):
    """ViT-B / ViT-L geometries (embed 768 / 1024, 24 blocks, DPT 128 / 256 wide, the chunked 768 / 1024-channel stride-2 conv),
    Depth-Anything V1's taps (the last four blocks) and the V2 metric head, all from the checkpoint's own shapes."""
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    sd = ODA.random_state_dict(620, grid=8, encoder=encoder)
    net = HipDepthAnythingV2(sd, "cuda:0", taps=taps, max_depth=max_depth)
    assert net.metric_depth == (max_depth > 0)
    x = _norm(torch.stack([synth_image(190 + i, 3, 70, 98) for i in range(2)]))
    ref = ODA.model_forward(sd, x, taps=taps, max_depth=max_depth)
    y = net(x.to("cuda:0")).cpu()
    assert y.shape == ref.shape == (2, 70, 98)
    span = float(ref.max() - ref.min())
    p = psnr(y / span, ref / span)
    rel = ((y - ref).pow(2).mean().sqrt() / ref.std()).item()
    assert float(ref.std()) > 1e-3 and p >= 50.0 and rel < 1.5e-2, (encoder, taps, max_depth, p, rel)


@pytest.mark.gpu
def test_hip_backbone_full_size_in_the_pipeline(hiplib):
    """392 x 686 (what batch_preprocess produces for 1080p) through BaseDepthModel.infer with the HIP backbone."""
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    from nunif_amd.iw3.base_depth_model import CallableDepthModel
    sd = ODA.random_state_dict(602)
    net = HipDepthAnythingV2(sd, "cuda:0")
    x = _norm(synth_image(95, 3, 392, 686)[None])
    ref = ODA.model_forward(sd, x)
    y = net(x.to("cuda:0")).cpu()
    span = float(ref.max() - ref.min())
    assert psnr(y / span, ref / span) >= 50.0, psnr(y / span, ref / span)
    model = CallableDepthModel(net)
    model.load(gpu=0)
    frame = synth_image(96, 3, 1080, 1920)
    d = model.infer(frame.to("cuda:0"), tta=False, edge_dilation=2)
    assert d.shape == (1, 392, 686) and torch.isfinite(d).all() and float(d.std()) > 0


@pytest.mark.gpu
def test_resident_weight_conv_is_bit_identical_to_the_ring_form(hiplib, monkeypatch):
    """conv_kernel<NT,MF,RES>: the LDS-resident persistent form (taken for >= 512 pixel groups, i.e. the large DPT-head maps
    at batch 2) issues the same MFMAs in the same order as the ring form."""
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    net = HipDepthAnythingV2(ODA.random_state_dict(602), "cuda:0")
    x = _norm(torch.stack([synth_image(95, 3, 392, 686), synth_image(97, 3, 392, 686)])).to("cuda:0")
    monkeypatch.setenv("NUNIF_CONV_RES", "1")
    a = net(x).cpu()
    monkeypatch.setenv("NUNIF_CONV_RES", "0")
    b = net(x).cpu()
    assert float(a.std()) > 0 and torch.equal(a, b)


@pytest.mark.gpu
def test_lds_staged_conv_is_bit_identical_to_the_gather_form(hiplib, monkeypatch):
    """conv3_lds_kernel (input halo staged in LDS once, resident or ringed weights) issues the same MFMAs in the same order as
    conv_kernel's per-tap gathers: the whole DPT head must not change by a bit, on maps that are not multiples of its 8 x 32
    patch (5 x 7 ... 80 x 112) and with 64- and 128-wide fusion maps (ViT-S / ViT-B).  The chunk-major form of the wide-input
    convs (conv3_lds_cm_kernel: layer3_rn / layer4_rn) contracts in a different order, so it is held to a tolerance instead."""
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    for enc, shape in (("vits", (2, 3, 70, 98)), ("vits", (1, 3, 154, 266)), ("vitb", (1, 3, 84, 126))):
        sd = ODA.random_state_dict(630, grid=8, encoder=enc)
        x = _norm(torch.stack([synth_image(290 + i, 3, shape[2], shape[3]) for i in range(shape[0])])).to("cuda:0")
        monkeypatch.setenv("NUNIF_CONV3_CM", "0")
        net = HipDepthAnythingV2(sd, "cuda:0")              # tap-major streams everywhere
        monkeypatch.setenv("NUNIF_CONV3_LDS", "1")
        a = net(x).cpu()
        monkeypatch.setenv("NUNIF_CONV3_LDS", "0")
        b = net(x).cpu()
        assert float(a.std()) > 0 and torch.equal(a, b), (enc, shape)
        monkeypatch.delenv("NUNIF_CONV3_CM")
        monkeypatch.setenv("NUNIF_CONV3_LDS", "1")
        c = HipDepthAnythingV2(sd, "cuda:0")(x).cpu()       # chunk-major layer{3,4}_rn
        span = float(a.max() - a.min())
        assert float((c - a).abs().max()) < 2e-3 * span, (enc, shape, float((c - a).abs().max()), span)


@pytest.mark.gpu
def test_fused_mlp_kernel_matches_the_two_linear_launches(hiplib):
    """ViT-S: fc1 + GELU + fc2 + residual in one kernel (depth_mlp.hip) against the two gemm launches it replaces, in both forms:
    the hidden-split pair of workgroups (the default while the grid fits the chip) and the one-workgroup form behind it.  Same
    fp16 operands and fp32 accumulation; the contraction order of fc2 differs (chained k order, two halves), so not bit-identical.
    The split form hands fp32 partials from one workgroup to another: every launch is repeated and must reproduce itself."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import psnr, synth_image
from oracle import depth_anything_v2 as ODA
from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
net = HipDepthAnythingV2(ODA.random_state_dict(601), "cuda:0")
for shape in ((4, 392, 686), (2, 56, 70), (1, 28, 42), (3, 266, 490)):
    x = ((torch.stack([synth_image(120 + i, 3, shape[1], shape[2]) for i in range(shape[0])]) - mean) / std).to("cuda:0")
    os.environ["NUNIF_DA_MLP"] = "0"
    a = net(x).clone()
    os.environ["NUNIF_DA_MLP"] = "1"
    outs = [net(x).clone() for _ in range(4)]
    span = float(a.max() - a.min())
    for b in outs:
        rel = ((a - b).pow(2).mean().sqrt() / a.std()).item()
        assert torch.isfinite(b).all() and psnr(a / span, b / span) >= 55.0 and rel < 3e-3, (shape, rel)
        assert torch.equal(b, outs[0]), shape
print("OK")
"""
    for split in ("1", "0"):
        env = dict(os.environ, NUNIF_DA_MLP_SPLIT=split)        # read once per process by the launcher
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "OK" in r.stdout, (split, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_reassemble_branches_on_side_streams_change_nothing(hiplib, monkeypatch):
    """The project -> resize -> layer_rn branch of a tap runs on a stream of its own beside the encoder layers that follow the tap
    (depth_anything.hip forward()): same kernels, same operands — the depth map must be EQUAL to the single-stream one, also when
    calls follow each other without a synchronisation in between (the branches of call n + 1 reuse the buffers of call n)."""
    from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2
    net = HipDepthAnythingV2(ODA.random_state_dict(601), "cuda:0")
    xs = [_norm(torch.stack([synth_image(140 + 7 * j + i, 3, 392, 686) for i in range(4)])).to("cuda:0") for j in range(3)]
    monkeypatch.setenv("NUNIF_DA_BRANCH_STREAMS", "0")
    ref = [net(x).clone() for x in xs]
    monkeypatch.setenv("NUNIF_DA_BRANCH_STREAMS", "1")
    for _ in range(3):
        outs = [net(x) .clone() for x in xs]                 # back to back, no synchronisation between the calls
        torch.cuda.synchronize()
        for a, b in zip(ref, outs):
            assert torch.equal(a, b), float((a - b).abs().max())
    small = _norm(torch.stack([synth_image(170, 3, 56, 70)])).to("cuda:0")
    monkeypatch.setenv("NUNIF_DA_BRANCH_STREAMS", "0")
    a = net(small).clone()
    monkeypatch.setenv("NUNIF_DA_BRANCH_STREAMS", "1")
    assert torch.equal(a, net(small))
