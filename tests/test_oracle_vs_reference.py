"""Cross-check of the oracle against the LIVE reference (build container only; skipped where /root/reference is
absent, e.g. on the GPU box — the committed golden fixtures cover that case)."""
import pytest
import torch

from oracle import refstub
from oracle import seam_blending as OS
from oracle import swin_unet as O

pytestmark = pytest.mark.skipif(not refstub.reference_available(), reason="reference tree not mounted")


@pytest.fixture(scope="module")
def ref():
    refstub.install()
    import waifu2x.models.swin_unet as ref_swin
    import nunif.utils.seam_blending as ref_seam
    return {"swin": ref_swin, "seam": ref_seam}


def test_param_counts_and_keys(ref):
    for cls, sf, n in ((ref["swin"].SwinUNet, 1, 3757431), (ref["swin"].SwinUNet2x, 2, 3758304),
                       (ref["swin"].SwinUNet4x, 4, 4302852)):
        m = cls()
        assert sum(p.numel() for p in m.parameters()) == n
        sd = O.random_state_dict(7, sf)
        msd = m.state_dict()
        assert set(msd) == set(sd)
        assert all(msd[k].shape == sd[k].shape for k in sd)
        assert torch.equal(msd["unet.swin1.block.1.attn.relative_position_index"],
                           sd["unet.swin1.block.1.attn.relative_position_index"])


def test_random_sizes_grid(ref):
    g = torch.Generator().manual_seed(5)
    SB = ref["seam"].SeamBlending
    for _ in range(200):
        h, w = (int(v) for v in torch.randint(1, 3000, (2,), generator=g))
        s, o, b, t = [(1, 8, 4, 64), (2, 16, 8, 256), (4, 32, 16, 112), (1, 28, 0, 256), (2, 36, 0, 128)][
            int(torch.randint(0, 5, (1,), generator=g))]
        assert SB.create_config((h, w), s, o, t, b) == OS.create_config(h, w, s, o, t, b)


def test_forward_2x_random_tile(ref):
    sd = O.random_state_dict(31, 2)
    m = ref["swin"].SwinUNet2x().eval()
    m.load_state_dict(sd)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    assert (m(x) - O.model_forward(sd, x)).abs().max().item() < 2e-4
