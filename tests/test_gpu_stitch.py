"""HIP tile gather + single-pass stitcher against the oracle (bit-exact), through the C ABI."""
import ctypes

import pytest
import torch

from oracle import seam_blending as OS

pytestmark = pytest.mark.gpu

CASES = [  # (H, W, scale, offset, blend, tile, C)
    (100, 130, 2, 16, 8, 64, 3),      # ragged, vector path (To=96, step 88)
    (75, 141, 1, 8, 4, 64, 3),        # 1x
    (61, 67, 4, 32, 16, 64, 3),       # 4x
    (90, 70, 1, 28, 0, 64, 3),        # cunet geometry: no blending
    (50, 77, 2, 36, 0, 64, 1),        # upcunet geometry, 1 channel
    (33, 45, 2, 15, 7, 63, 3),        # odd everything -> scalar path, ceil() offsets, overlapping no-vec
    (1, 1, 2, 16, 8, 64, 3),          # minimum size
    (40, 40, 2, 17, 0, 64, 2),        # blend 0 with overlapping tiles (offset not divisible by scale): last wins
]


def _fake_model(scale, offset, seed_scale=1.0):
    """Deterministic, position- and content-dependent stand-in for a network (exact in fp32 on both sides)."""
    def fn(mb):
        up = torch.nn.functional.interpolate(mb, scale_factor=scale, mode="nearest")
        n = up.shape[-1]
        up = up[:, :, offset:n - offset, offset:n - offset]
        ramp = torch.linspace(0, 0.5, up.shape[-1], device=mb.device)
        return (up * 0.75 + ramp[None, None, :, None] * ramp[None, None, None, :]) * seed_scale
    return fn


@pytest.mark.parametrize("H,W,scale,offset,blend,tile,C", CASES)
def test_gather_and_stitch_bit_exact(hiplib, H, W, scale, offset, blend, tile, C):
    from nunif_amd.nunif.utils.seam_blending import SeamBlending
    dev = torch.device("cuda:0")
    x = torch.rand(C, H, W, generator=torch.Generator().manual_seed(H * 1000 + W))
    fn = _fake_model(scale, offset)
    expect = OS.tiled_render(x, fn, scale, offset, blend, tile, 4)

    sb = SeamBlending((C, H, W), scale, offset, tile, blend)
    cfg = OS.create_config(H, W, scale, offset, tile, blend)
    n = sb.h_blocks * sb.w_blocks
    assert (sb.h_blocks, sb.w_blocks) == (cfg["h_blocks"], cfg["w_blocks"])
    xd = x.to(dev)
    tiles = torch.empty((n, C, tile, tile), device=dev)
    sb.gather(xd, 0, n, tiles)
    xp = torch.nn.functional.pad(x[None], cfg["pad"], mode="replicate")[0]
    step = cfg["input_tile_step"]
    ref_tiles = torch.stack([xp[:, i * step:i * step + tile, j * step:j * step + tile]
                             for i in range(cfg["h_blocks"]) for j in range(cfg["w_blocks"])])
    assert torch.equal(tiles.cpu(), ref_tiles), "tile gather (replicate pad + slice) differs"
    # partial gather ranges
    part = torch.empty((1, C, tile, tile), device=dev)
    sb.gather(xd, n - 1, 1, part)
    assert torch.equal(part.cpu()[0], ref_tiles[-1])

    z = fn(ref_tiles)                      # CPU fp32, identical inputs for both stitchers
    store = sb._store(dev)
    store.copy_(z.to(dev))
    out = sb.get_output().cpu()
    assert out.shape == expect.shape
    assert torch.equal(out, expect), f"stitch differs: max abs {(out - expect).abs().max().item():.3e}"


def test_tiled_render_generic_model_path(hiplib):
    """SeamBlending.tiled_render with an arbitrary torch callable as the model (contract B1/B2)."""
    from nunif_amd.nunif.utils.render import tiled_render

    class Fake(torch.nn.Module):
        i2i_scale, i2i_offset, i2i_blend_size, i2i_default_tile_size, i2i_default_batch_size = 2, 16, 8, 64, 4

        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.fn = _fake_model(2, 16)

        def find_valid_tile_size(self, t):
            return t or 64

        def forward(self, x):
            return self.fn(x)

    m = Fake().to("cuda:0").eval()
    x = torch.rand(3, 120, 200, generator=torch.Generator().manual_seed(1))
    out = tiled_render(x, m, tile_size=64, batch_size=3)
    expect = OS.tiled_render(x, _fake_model(2, 16), 2, 16, 8, 64, 4)
    assert out.device.type == "cuda"
    assert (out.cpu() - expect).abs().max().item() < 1e-6


def test_stitch_rejects_null(hiplib):
    from nunif_amd import _hip
    g = _hip.tile_grid(10, 10, 2, 16, 64, 8)
    rc = hiplib.nunif_hip_stitch_tiles(None, None, ctypes.byref(g), 3, None)
    assert rc == -1 and b"NULL" in hiplib.nunif_hip_last_error()
