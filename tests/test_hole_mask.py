"""sbs.mask_mlbw_l2 (MLBW with the hole-logit channel), ``postprocess_hole_mask`` and ``nonwarp_mask``: oracle vs the
reference fixture (CPU), HIP engine vs fixture (GPU).  Masks produced from the SAME logits must be bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, psnr, sd_checksum
from oracle import mlbw as OM
from oracle import row_flow_v3 as ORF

MASKS = {"mask_same": ((58, 104), 0.15, 0, 0), "mask_up": ((116, 208), 0.15, 0, 0), "mask_dil": ((116, 208), 0.15, 1, 2),
         "mask_odd": ((131, 259), 0.3, 0, 1)}


@pytest.fixture(scope="module")
def g():
    raw = np.load(os.path.join(GOLDEN, "hole_mask.npz"))
    out = {}
    for k in raw.files:
        v = raw[k]
        if k in MASKS or k == "nonwarp":
            shape = (2, 1) + (MASKS[k][0] if k in MASKS else (116, 208))
            v = np.unpackbits(v)[:int(np.prod(shape))].reshape(shape).astype(bool)
        out[k] = torch.from_numpy(v)
    return out


def _sd():
    return OM.random_state_dict(431, 2, False, hole_mask=True)


def test_oracle_matches_reference_fixture(g):
    sd = _sd()
    assert sd["lv1_out.1.weight"].shape[0] == 5
    assert sd_checksum(sd) == pytest.approx(float(g["sdsum"]), rel=1e-12)
    depth, c = g["depth"], g["c"]
    d, w, lg = OM.delta_forward(sd, ORF.make_input(depth[:1], 2.0, 0.5, 104), 2)
    assert lg.shape == g["logits"].shape == (1, 1, 58, 104)
    assert (d - g["delta"]).abs().max().item() < 5e-5 and (lg - g["logits"]).abs().max().item() < 5e-5
    for tag, shift in (("l", -1), ("r", 1)):
        z, lgs = OM.apply_divergence_nn_delta_weight(sd, c, depth, 2.0, 0.5, shift, 2, return_mask=True)
        assert (lgs - g["logits_" + tag]).abs().max().item() < 5e-5
        assert (z - g["z_" + tag].float()).abs().max().item() < 1e-3            # fixture stored as fp16
    # the post-processing is bit-exact on the reference's own logits
    for k, (size, thr, ni, no) in MASKS.items():
        assert torch.equal(OM.postprocess_hole_mask(g["logits_r"], size, thr, ni, no), g[k]), k
    fill = OM.apply_divergence_nn_delta_weight(sd, c, depth, 2.0, 0.5, 1, 2)
    assert ((fill - g["fill_r"].float()).abs() > 1e-3).float().mean().item() < 1e-3
    _, mask = OM.nonwarp_mask(sd, c, depth, 4.0, 0.5, 2, 0.15, 1, 1)
    assert (mask != g["nonwarp"]).float().mean().item() < 1e-3


@pytest.mark.gpu
def test_hip_postprocess_bit_exact(hiplib, g):
    from nunif_amd.iw3.backward_warp import postprocess_hole_mask
    lg = g["logits_r"].to("cuda:0")
    for k, (size, thr, ni, no) in MASKS.items():
        m = postprocess_hole_mask(lg, size, thr, inner_dilation=ni, outer_dilation=no)
        assert m.dtype == torch.bool and tuple(m.shape) == (2, 1) + size
        assert torch.equal(m.cpu(), g[k]), (k, (m.cpu() != g[k]).sum().item())
    with pytest.raises(RuntimeError):
        postprocess_hole_mask(g["logits_r"], (58, 104), 0.15)                   # CPU tensor: no fallback


@pytest.mark.gpu
def test_hip_mask_mlbw(hiplib, g):
    from nunif_amd.nunif.models import create_model
    from nunif_amd.iw3 import models  # noqa: F401
    from nunif_amd.iw3.backward_warp import apply_divergence_nn_delta_weight, make_input_tensor, nonwarp_mask
    m = create_model("sbs.mask_mlbw_l2").eval()
    m.load_state_dict(_sd(), strict=True)
    m = m.to("cuda:0")
    m.delta_output = True
    assert m.hole_mask and m.num_layers == 2
    depth, c = g["depth"].to("cuda:0"), g["c"].to("cuda:0")
    d, w, lg = m(torch.stack([make_input_tensor(None, depth[0], 2.0, 0.5, 104)]))
    el = (lg.cpu() - g["logits"]).abs()
    assert el.max().item() < 8e-2 and el.mean().item() < 6e-3, (el.max().item(), el.mean().item())
    for tag, shift in (("l", -1), ("r", 1)):
        z, lgs = apply_divergence_nn_delta_weight(m, c, depth, 2.0, 0.5, steps=1, shift=shift, return_mask=True)
        assert psnr(z.cpu(), g["z_" + tag].float()) >= 50.0
        e = (lgs.cpu() - g["logits_" + tag]).abs()
        assert e.max().item() < 8e-2 and e.mean().item() < 6e-3, (tag, e.max().item(), e.mean().item())
        fill = apply_divergence_nn_delta_weight(m, c, depth, 2.0, 0.5, steps=1, shift=shift)
        bad = ((fill.cpu() - g["fill_" + tag].float()).abs() > 2e-2).float().mean().item()
        assert bad < 0.02, (tag, bad)            # pixels whose logit sits within the fp16 error of the threshold may flip
    _, mask = nonwarp_mask(m, c, depth, 4.0, 0.5, threshold=0.15, inner_dilation=1, outer_dilation=1)
    assert mask.dtype == torch.bool and (mask.cpu() != g["nonwarp"]).float().mean().item() < 0.02
