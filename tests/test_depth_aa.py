"""iw3.depth_aa: oracle vs the reference fixture (CPU), HIP engine vs fixture / oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr, sd_checksum
from oracle import depth_aa as ODA
from oracle.forward_warp import synth_depth


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "depth_aa.npz")).items()}


def test_oracle_matches_reference_fixture(g):
    sd = ODA.random_state_dict(501)
    assert sd_checksum(sd) == pytest.approx(float(g["sdsum"]), rel=1e-12)
    assert (ODA.forward(sd, g["x"]) - g["y"]).abs().max().item() < 1e-5
    assert (ODA.forward(sd, g["x"], clamp=False) - g["y_noclamp"]).abs().max().item() < 1e-5
    assert (ODA.infer(sd, g["xi"]) - g["y_infer"]).abs().max().item() < 5e-5
    change = (g["y_noclamp"] - g["x"]).abs()
    assert change.mean().item() > 5e-3            # the fixture's net really edits the depth


@pytest.mark.gpu
def test_hip_depth_aa(hiplib, g):
    from nunif_amd.iw3.models import DepthAA
    sd = ODA.random_state_dict(501)
    m = DepthAA().eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    assert (m.name, m.i2i_scale, m.i2i_offset) == ("iw3.depth_aa", 1, 0)
    x = g["x"].to("cuda:0")
    y = m(x)
    assert y.shape == x.shape and float(y.min()) >= 0 and float(y.max()) <= 1
    # the quantity that matters is the EDIT the net makes: compare it, not just the (dominant) pass-through
    edit_ref, edit = g["y_noclamp"] - g["x"], m(x, clamp=False).cpu() - g["x"]
    rel = (edit - edit_ref).pow(2).mean().sqrt() / edit_ref.pow(2).mean().sqrt()
    assert rel.item() < 2e-2, rel.item()
    assert psnr(y.cpu(), g["y"]) >= 50.0 and psnr(m(x, clamp=False).cpu(), g["y_noclamp"]) >= 50.0
    yi = m.infer(g["xi"].to("cuda:0")).cpu()
    span = float(g["xi"].max() - g["xi"].min())
    assert psnr(yi / span, g["y_infer"] / span) >= 50.0
    assert torch.equal(m(x[1:2]), y[1:2])                     # batch independent
    # DepthAnything's output size
    d = synth_depth(9, 1, 392, 686, "smooth_edges")
    z = m(d.to("cuda:0")).cpu()
    assert psnr(z, ODA.forward(sd, d)) >= 50.0
