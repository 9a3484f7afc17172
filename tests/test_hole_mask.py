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


def _morph_golden():
    import os
    import numpy as np
    from conftest import GOLDEN
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "morph.npz")).items()}


def test_oracle_mask_morphology_matches_reference():
    """oracle.dilation mask morphology == the reference's iw3/dilation.py outputs (tests/golden/morph.npz), exactly."""
    from oracle import dilation as OD
    g = _morph_golden()
    m = g["mask"]
    mf = m.float()
    assert torch.equal(OD.dilate(mf), g["dilate"]) and torch.equal(OD.erode(mf), g["erode"])
    assert torch.equal(OD.closing(m), g["closing2"]) and torch.equal(OD.closing(m, n_iter=1), g["closing1"])
    assert torch.equal(OD.mask_closing(m), g["mask_closing2"])
    assert torch.equal(OD.dilate_outer(m, 3), g["outer3"]) and torch.equal(OD.dilate_inner(m, 2), g["inner2"])
    assert torch.equal(OD.dilate_outer(m, 4, base_width=74), g["outer_bw"])


@pytest.mark.gpu
def test_hip_mask_morphology_and_named_mappers(hiplib):
    """nunif_amd.iw3.dilation mask helpers (bit-exact) and nunif_amd.iw3.mapper named mappers vs the reference fixture."""
    from nunif_amd.iw3 import dilation as D
    from nunif_amd.iw3 import mapper as M
    g = _morph_golden()
    m = g["mask"].to("cuda:0")
    mf = m.float()
    assert torch.equal(D.dilate(mf).cpu(), g["dilate"]) and torch.equal(D.erode(mf).cpu(), g["erode"])
    assert torch.equal(D.closing(m).cpu(), g["closing2"]) and torch.equal(D.closing(m, n_iter=1).cpu(), g["closing1"])
    assert torch.equal(D.mask_closing(m).cpu(), g["mask_closing2"])
    o3, i2 = D.dilate_outer(m, 3), D.dilate_inner(m, 2)
    assert o3.dtype == torch.bool and torch.equal(o3.cpu(), g["outer3"]) and torch.equal(i2.cpu(), g["inner2"])
    assert torch.equal(D.dilate_outer(m, 4, base_width=74).cpu(), g["outer_bw"])
    assert D.dilate_outer(m, 0) is m
    x = g["x"].to("cuda:0")
    for name, got in (("softplus01", M.softplus01(x, 0.343, 12)), ("inv_softplus01", M.inv_softplus01(x, -0.002102, 7.8788)),
                      ("softplus01_legacy", M.softplus01_legacy(x, 6)), ("distance_to_disparity", M.distance_to_disparity(x, 0.6)),
                      ("shift_relative_depth", M.shift_relative_depth(x, 1.4))):
        assert (got.cpu() - g[name]).abs().max().item() < 2e-6, name


def test_oracle_forward_nonwarp_mask_matches_reference():
    from oracle import forward_warp as OF
    g = _morph_golden()
    for view in ("right", "left"):
        cc, mask = OF.nonwarp_mask(g["fw_c"], g["fw_depth"], 16.0, 0.5, view=view)
        assert torch.equal(cc, g["fw_c"]) and torch.equal(mask, g["fw_mask_" + view]), view
        assert 0.0 < float(mask.mean()) < 0.5


@pytest.mark.gpu
def test_hip_forward_nonwarp_mask_and_backward_warp(hiplib):
    """iw3.forward_warp.nonwarp_mask (bit-exact, two forward-warp launches) and the generic backward_warp / make_grid /
    pad_delta_y entry points vs the reference fixture."""
    from nunif_amd.iw3 import backward_warp as B
    from nunif_amd.iw3 import forward_warp as FW
    g = _morph_golden()
    c, depth = g["fw_c"].to("cuda:0"), g["fw_depth"].to("cuda:0")
    for view in ("right", "left"):
        cc, mask = FW.nonwarp_mask(c, depth, 16.0, 0.5, view=view)
        assert torch.equal(cc.cpu(), g["fw_c"])
        assert torch.equal(mask.cpu(), g["fw_mask_" + view]), view
    delta_x = g["bw_delta_x"].to("cuda:0")
    grid = B.make_grid(1, 32, 20, "cuda:0")
    assert grid.shape == (1, 2, 20, 32)
    out = B.backward_warp(c, grid, B.pad_delta_y(delta_x), 1.0 / (64 // 2 - 1)).cpu()
    assert (out - g["bw_out"]).abs().max().item() < 1e-5
    with pytest.raises(NotImplementedError):
        B.backward_warp(c, grid.clone(), B.pad_delta_y(delta_x), 1.0)
