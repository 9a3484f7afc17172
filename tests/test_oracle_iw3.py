"""iw3 oracles against the committed reference outputs (tests/golden/iw3.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import backward_warp as OB
from oracle import depth_pre as OP
from oracle import dilation as OD
from oracle import forward_warp as OF


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "iw3.npz")).items()}


def test_forward_warp_bit_exact(g):
    for tag, depth in (("full", g["depth"]), ("small", g["depth_small"])):
        out = OF.forward_warp(g["c"], depth, 40.0, 0.5, fill=True, synthetic_view="both", return_mask=True,
                              width_base=False)
        for got, key in zip(out, ("left", "right", "lmask", "rmask")):
            assert torch.equal(got, g[f"fw_{tag}_{key}"]), (tag, key)
    le, ri = OF.forward_warp(g["c"], g["depth"], 12.0, 0.2, fill=False)
    assert torch.equal(le, g["fw_nofill_left"]) and torch.equal(ri, g["fw_nofill_right"])
    le, ri = OF.forward_warp(g["c"], g["depth"], 8.0, 0.5, fill=True, synthetic_view="right")
    assert le is g["c"] and torch.equal(ri, g["fw_right_only"])
    assert float(g["fw_full_lmask"].sum()) > 0, "fixture has no holes: the hole logic would be untested"
    assert float((g["fw_full_lmask"] == 0.5).sum()) + float((g["fw_full_rmask"] == 0.5).sum()) > 0, \
        "fixture has no layered holes"


def test_forward_warp_constant_depth_is_a_pure_shift():
    """Known answer: constant depth d, convergence conv -> every pixel moves by s=(d-conv)*shift_size; with an
    integer s the left eye is the image shifted right by s, holes on the left filled from... the zero inflow."""
    c = torch.rand(1, 3, 8, 50)
    d = torch.full((1, 1, 8, 50), 1.0)
    le, ri = OF.forward_warp(c, d, 8.0, 0.0, fill=False)         # shift_size = 8*0.01*50*0.5 = 2 px
    assert torch.allclose(le[..., 2:], c[..., :-2], atol=1e-5)   # left eye: content moves right
    assert torch.allclose(ri[..., :-2], c[..., 2:], atol=1e-5)
    # columns 0-1 of the left eye receive the replicate-padded border column
    assert torch.allclose(le[..., :2], c[..., :1].expand(1, 3, 8, 2), atol=1e-5)


def test_shift_fill_and_layered_holes_closed_forms():
    x = torch.tensor([[[[-1.0, 3.0, -1.0, -1.0, 5.0, -2.0]]]])
    assert OF.shift_fill(x, -1).flatten().tolist() == [0.0, 3.0, 3.0, 3.0, 5.0, 5.0]
    assert OF.shift_fill(x, 1).flatten().tolist() == [3.0, 3.0, 5.0, 5.0, 5.0, 0.0]
    long = torch.full((1, 1, 1, 150), -1.0)
    long[..., 0] = 0.25
    filled = OF.shift_fill(long, -1)
    assert filled[..., :101].eq(0.25).all() and filled[..., 101:].eq(-1.0).all()      # 100-step cap
    idx = torch.tensor([[[[0.0, 4.0, 2.0, 3.0, 1.0, 5.0]]]])
    side = torch.zeros(1, 3, 1, 6)
    s2, i2 = OF.fix_layered_holes(side, idx, 1)
    assert i2.flatten().tolist() == [0.0, 1.0, 1.0, 1.0, 1.0, 5.0]
    assert s2[0, 0, 0].tolist() == [0.0, -2.0, -2.0, -2.0, 0.0, 0.0]


def test_grid_sample_dilate_preprocess_bit_exact(g):
    le, ri = OB.grid_sample_warp(g["c"], g["depth"], 2.5, 0.3)
    assert torch.equal(le, g["gs_left"]) and torch.equal(ri, g["gs_right"])
    le, ri = OB.grid_sample_warp(g["c"], g["depth_small"], 2.5, 0.3)
    assert torch.equal(le, g["gs_small_left"]) and torch.equal(ri, g["gs_small_right"])
    assert torch.equal(OD.dilate_edge(g["raw_depth"], [2, 1]), g["dilate_2_1"])
    assert torch.equal(OD.dilate_edge(g["raw_depth"], [1, 3]), g["dilate_1_3"])
    assert torch.equal(OD.dilate_edge(g["raw_depth"], 2), g["dilate_2"])
    assert torch.equal(OP.batch_preprocess(g["pre_in"], lower_bound=56), g["pre_out"])
    assert OP.preprocess_size(1080, 1920) == (392, 686) and OP.preprocess_size(1920, 1080) == (686, 392)
    assert OP.preprocess_size(100, 1000) == (392, 1568)        # aspect cap 4


def test_dilate_edge_known_answers():
    assert OD.parse(None) == (0, 0) and OD.parse(3) == (3, 3) and OD.parse([2, 1]) == (2, 1) and OD.parse((4,)) == (4, 4)
    with pytest.raises(ValueError):
        OD.parse("2")
    flat = torch.full((1, 1, 12, 12), 0.3)
    assert torch.allclose(OD.dilate_edge(flat, 2), flat)                 # no edges -> unchanged
    step = torch.zeros(1, 1, 16, 16)
    step[..., 8:] = 1.0
    out = OD.dilate_edge(step, [2, 0])
    assert out[0, 0, 8, 7] > 0.2 and out[0, 0, 8, 2] == 0 and out[0, 0, 8, 12] == 1     # the near side grows
