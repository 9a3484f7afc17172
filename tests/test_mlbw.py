"""sbs.mlbw (multi-layer backward warp): oracle vs the reference fixture (CPU), HIP engine vs fixture / oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr, sd_checksum
from oracle import mlbw as OM
from oracle import row_flow_v3 as ORF

VARIANTS = (("l2", 2, False, 2), ("l4", 4, False, 1), ("l2s", 2, True, 1))


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "mlbw.npz")).items()}


def _sd(L, small):
    return OM.random_state_dict(400 + L + (10 if small else 0), L, small)


@pytest.mark.parametrize("tag,L,small,nb", VARIANTS)
def test_oracle_matches_reference_fixture(g, tag, L, small, nb):
    sd = _sd(L, small)
    assert sd_checksum(sd) == pytest.approx(float(g[tag + "_sdsum"]), rel=1e-12)
    depth, c = g["depth"], g["c"]
    d, w = OM.delta_forward(sd, ORF.make_input(depth[:1], 2.0, 0.5, 104), L)
    assert d.shape == g[tag + "_delta"].shape == (1, L, 58, 104)
    assert (d - g[tag + "_delta"]).abs().max().item() < 5e-5 and (w - g[tag + "_weight"]).abs().max().item() < 1e-5
    assert float(w.sum(dim=1).sub(1).abs().max()) < 1e-5 and float(g[tag + "_weight"].max()) > 0.9     # layers really compete
    left, right = OM.apply_divergence_nn_LR(sd, c[:nb], depth[:nb], 2.0, 0.5, L)
    assert (left - g[tag + "_left"]).abs().max().item() < 2e-4 and (right - g[tag + "_right"]).abs().max().item() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("tag,L,small,nb", VARIANTS)
def test_hip_mlbw(hiplib, g, tag, L, small, nb):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401  (registers the factories)
    from nunif_amd.iw3.backward_warp import apply_divergence_nn_LR, make_input_tensor
    sd = _sd(L, small)
    m = create_model("sbs.mlbw_" + tag).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    m.delta_output = True
    assert (m.name, m.num_layers, m.i2i_offset) == ("sbs.mlbw", L, 32)
    depth, c = g["depth"].to("cuda:0"), g["c"].to("cuda:0")
    d, w = m(torch.stack([make_input_tensor(None, depth[0], 2.0, 0.5, 104)]))
    ed, ew = (d.cpu() - g[tag + "_delta"]).abs(), (w.cpu() - g[tag + "_weight"]).abs()
    assert ed.max().item() < 5e-2 and ed.mean().item() < 4e-3, (ed.max().item(), ed.mean().item())
    assert ew.max().item() < 3e-2 and ew.mean().item() < 2e-3, (ew.max().item(), ew.mean().item())
    left, right = apply_divergence_nn_LR(m, c[:nb], depth[:nb], 2.0, 0.5, steps=1, synthetic_view="both")
    pl, pr = psnr(left.cpu(), g[tag + "_left"]), psnr(right.cpu(), g[tag + "_right"])
    assert pl >= 50.0 and pr >= 50.0, (pl, pr)
    l2, _ = apply_divergence_nn_LR(m, c[:1], depth[:1], 2.0, 0.5, steps=1, synthetic_view="both")
    assert torch.equal(l2, left[:1])
