"""swin_unet HIP engine against the oracle and the reference-generated golden fixtures.

Tolerance: float PSNR 10*log10(1/(mse+1e-6)) >= 50 dB on the [0,1] output (BASELINE.json north_star); the engine
stores activations in fp16 and accumulates in fp32, the oracle is fp32 end to end.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import psnr, sd_checksum, synth_image
from oracle import seam_blending as OS
from oracle import swin_unet as O

pytestmark = pytest.mark.gpu
PSNR_MIN = 50.0
NAMES = {1: "waifu2x.swin_unet_1x", 2: "waifu2x.swin_unet_2x", 4: "waifu2x.swin_unet_4x"}


def make_model(sf, seed):
    from nunif_amd.waifu2x.models import swin_unet as M
    cls = {1: M.SwinUNet, 2: M.SwinUNet2x, 4: M.SwinUNet4x}[sf]
    sd = O.random_state_dict(seed, sf)
    m = cls().eval()
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0"), sd


def read_taps(engine):
    from nunif_amd import _hip
    lib = _hip.lib()
    taps = {}
    i = 0
    while True:
        name = ctypes.create_string_buffer(64)
        nbytes = ctypes.c_int64(0)
        rc = lib.nunif_hip_swin_unet_get_tap(engine.handle, i, name, 64, None, 0, ctypes.byref(nbytes))
        if rc == 1:
            break
        _hip.check(rc)
        buf = np.empty(nbytes.value // 2, dtype=np.float16)
        _hip.check(lib.nunif_hip_swin_unet_get_tap(engine.handle, i, name, 64, buf.ctypes.data_as(ctypes.c_void_p),
                                                   nbytes.value, ctypes.byref(nbytes)))
        taps[name.value.decode()] = torch.from_numpy(buf.astype(np.float32))
        i += 1
    return taps


def _stagewise(hiplib, capsys, m, sd, sf, min_taps):
    from nunif_amd import _hip
    x = torch.stack([synth_image(21, 3, 64, 64), synth_image(22, 3, 64, 64)])
    ref_taps = {}
    y_ref = torch.clamp(O.unet_forward(sd, x, sf, taps=ref_taps), 0, 1)
    eng = m.engine()
    _hip.check(hiplib.nunif_hip_swin_unet_debug_taps(eng.handle, 1))
    y = m(x.to("cuda:0")).cpu()
    taps = read_taps(eng)
    _hip.check(hiplib.nunif_hip_swin_unet_debug_taps(eng.handle, 0))
    assert set(taps) <= set(ref_taps) and len(taps) >= min_taps    # fused kernels expose fewer intermediates
    report, worst = [], 0.0
    for name, ref in ref_taps.items():
        if name not in taps:
            continue
        got = taps[name].reshape(ref.shape)
        rel = ((got - ref).pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-12)).item()
        report.append(f"{name:18s} rel_rms={rel:.2e} ref_rms={ref.pow(2).mean().sqrt().item():.3f}")
        worst = max(worst, rel)
    with capsys.disabled():
        print("\n" + "\n".join(report))
        print(f"final PSNR {psnr(y, y_ref):.2f} dB")
    assert worst < 2e-2, "a stage deviates by more than fp16 noise:\n" + "\n".join(report)
    assert psnr(y, y_ref) >= PSNR_MIN


def test_stagewise_taps_2x(hiplib, capsys):
    """Every intermediate of the 2x net against the oracle (relative RMS error per stage) — localises a wrong
    kernel to its stage.  Tile 64: level maps 48 / 24 / 12 (the 12x12 map has 2x2 windows: shift + mask active)."""
    m, sd = make_model(2, 102)
    _stagewise(hiplib, capsys, m, sd, 2, 31)


def test_stagewise_taps_4x(hiplib, capsys):
    """The 4x net: proj2 on the skip, a C = 192 TOP level (swin5 on the C = 192 kernels, ToImage as its own GEMM)."""
    m, sd = make_model(4, 104)
    _stagewise(hiplib, capsys, m, sd, 4, 31)


def test_stagewise_taps_4xl(hiplib, capsys):
    """swin_unet_4xl (base_dim 192, 12 heads, LayerNormNoBias): every block on the generic path (layernorm + GEMMs + the
    fused-qkv-map attention), K = 1536 PatchDown as two passes, unfused stem."""
    from nunif_amd.nunif.models import create_model
    m = create_model("waifu2x.swin_unet_4xl").eval()
    sd = O.random_state_dict(204, 4, base_dim=192, layer_norm=True)
    m.load_state_dict(sd, strict=True)
    _stagewise(hiplib, capsys, m.to("cuda:0"), sd, 4, 31)


@pytest.mark.parametrize("sf,tag", [(1, "1x"), (2, "2x"), (4, "4x")])
def test_forward_matches_golden_and_oracle(hiplib, golden_swin, sf, tag):
    m, sd = make_model(sf, 100 + sf)
    x = torch.from_numpy(golden_swin["x"])
    y = m(x.to("cuda:0")).cpu()
    ref = torch.from_numpy(golden_swin["y_" + tag])          # the reference's own output
    assert y.shape == ref.shape and y.dtype == torch.float32
    assert float(y.min()) >= 0.0 and float(y.max()) <= 1.0
    assert psnr(y, ref) >= PSNR_MIN, f"PSNR vs reference fixture {psnr(y, ref):.2f} dB"
    assert psnr(y, O.model_forward(sd, x, NAMES[sf])) >= PSNR_MIN


@pytest.mark.parametrize("sf,tag", [(1, "1x"), (2, "2x"), (4, "4x")])
def test_forward_matches_the_hf_backed_reference_fixture(hiplib, golden_swin_hf, sf, tag):
    """The same nets against ``swin_unet_hf.npz``: the reference's own ``SwinUNetBase`` over HuggingFace's ``SwinLayer``
    (``oracle/hf_pin.py``) — no line of ``oracle/tv_swin_block.py`` took part in producing the expected output."""
    m, sd = make_model(sf, 100 + sf)
    assert sd_checksum(sd) == pytest.approx(float(golden_swin_hf["sdsum_" + tag]), rel=1e-12)
    y = m(torch.from_numpy(golden_swin_hf["x"]).to("cuda:0")).cpu()
    ref = torch.from_numpy(golden_swin_hf["y_" + tag])
    assert y.shape == ref.shape and 0.05 < ref.std().item() < 0.45
    assert psnr(y, ref) >= PSNR_MIN, f"PSNR vs HF-backed reference fixture {psnr(y, ref):.2f} dB"
    if sf == 2:
        y = m(torch.from_numpy(golden_swin_hf["x_112"]).to("cuda:0")).cpu()
        assert psnr(y, torch.from_numpy(golden_swin_hf["y_2x_112"])) >= PSNR_MIN


@pytest.mark.parametrize("env", [{"NUNIF_BLOCK96": "1"}, {"NUNIF_ATT_WM": "0"}, {"NUNIF_BLOCK96": "1", "NUNIF_ATT_WM": "0"}])
@pytest.mark.parametrize("sf,tag", [(1, "1x"), (2, "2x"), (4, "4x")])
def test_every_kernel_switch_of_the_engine_matches_the_reference(hiplib, golden_swin, golden_swin_hf, monkeypatch, env, sf, tag):
    """The two switches the swin engine still reads at create time, each against the reference fixtures (>= 50 dB):
    NUNIF_BLOCK96=1 = one kernel per C = 96 block (swin_block96.hip, incl. its fused image head on the 1x / 2x nets) instead of
    attention + tail; NUNIF_ATT_WM=0 = pixel-major att map.  Also a whole tiled render (ragged frame, 2 x 3 tiles)."""
    from nunif_amd.nunif.utils.render import tiled_render
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m, sd = make_model(sf, 100 + sf)
    y = m(torch.from_numpy(golden_swin["x"]).to("cuda:0")).cpu()
    assert psnr(y, torch.from_numpy(golden_swin["y_" + tag])) >= PSNR_MIN
    assert psnr(y, torch.from_numpy(golden_swin_hf["y_" + tag])) >= PSNR_MIN
    if sf == 1:
        # the input that exposed a softmax stabiliser reduced over one lane group only (common.h row_group_max): the 1x net with
        # seed 303 on a 70 x 90 image has shifted windows whose logits differ by > 16 log2-units between lane groups
        m1, sd1 = make_model(1, 303)
        img = synth_image(71, 3, 70, 90)
        ref = OS.tiled_render(img, lambda mb: O.model_forward(sd1, mb, NAMES[1]), 1, 8, 4, 64, 4)
        out = tiled_render(img.to("cuda:0"), m1, tile_size=64, batch_size=4).cpu()
        assert torch.isfinite(out).all() and psnr(out, ref) >= PSNR_MIN, psnr(out, ref)
    if sf == 2:
        m2, _ = make_model(2, 102)
        out = tiled_render(torch.from_numpy(golden_swin["img"]).to("cuda:0"), m2, tile_size=64, batch_size=4).cpu()
        assert psnr(out, torch.from_numpy(golden_swin["render_2x_t64_b4"])) >= PSNR_MIN
        # 1080p-sized level-1 maps: 9 windows per wave and the 4-window remainder tile of swin_block96 at its real occupancy
        x = torch.stack([synth_image(40 + i, 3, 256, 256) for i in range(3)]).to("cuda:0")
        monkeypatch.delenv("NUNIF_BLOCK96", raising=False)
        monkeypatch.delenv("NUNIF_ATT_WM", raising=False)
        base, _ = make_model(2, 102)
        assert psnr(m2(x).cpu(), base(x).cpu()) >= 55.0


def test_downscaled_4x_to_2x_and_1x(hiplib, golden_swin):
    """SwinUNet4x.to_2x()/to_1x() (the fallback when scale2x.pth is absent, waifu2x/utils.py:139-144)."""
    m, _ = make_model(4, 104)
    x = torch.from_numpy(golden_swin["x"]).to("cuda:0")
    for net, key, scale, offset, blend in ((m.to_2x().eval(), "y_4x_to2x", 2, 16, 8), (m.to_1x().eval(), "y_4x_to1x", 1, 8, 16)):
        assert (net.i2i_scale, net.i2i_offset, net.i2i_blend_size) == (scale, offset, blend)
        y = net(x).cpu()
        ref = torch.from_numpy(golden_swin[key])
        assert y.shape == ref.shape and psnr(y, ref) >= PSNR_MIN, psnr(y, ref)


def test_4x_pre_antialias(hiplib):
    """SwinUNet4x(pre_antialias=True): every tile goes through bicubic x2 up / x2 down before the net (swin_unet.py
    :252-258,281-282); the fused whole-frame render is bypassed, tiled_render loops over tiles."""
    import torch.nn.functional as F
    from nunif_amd.waifu2x.models import swin_unet as M
    from nunif_amd.nunif.utils.render import tiled_render
    sd = O.random_state_dict(104, 4)
    m = M.SwinUNet4x(pre_antialias=True).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    assert not hasattr(m, "render_frame") and hasattr(M.SwinUNet4x().eval(), "render_frame")
    x = synth_image(81, 3, 64, 64)[None]
    xa = F.interpolate(F.interpolate(x, size=(128, 128), mode="bicubic", align_corners=False, antialias=True),
                       size=(64, 64), mode="bicubic", align_corners=False, antialias=True)
    ref = O.model_forward(sd, xa, NAMES[4])
    y = m(x.to("cuda:0")).cpu()
    assert y.shape == ref.shape and psnr(y, ref) >= PSNR_MIN, psnr(y, ref)
    img = synth_image(82, 3, 80, 100)
    out = tiled_render(img, m, tile_size=64, batch_size=3)
    assert out.shape == (3, 320, 400) and float(out.min()) >= 0 and float(out.max()) <= 1
    assert torch.equal(out, tiled_render(img, m, tile_size=64, batch_size=5))


def test_forward_112_batch3_and_half_input(hiplib, golden_swin):
    m, sd = make_model(2, 102)
    x = torch.from_numpy(golden_swin["x_112"])
    ref = torch.from_numpy(golden_swin["y_2x_112"])
    xb = torch.cat([x, x.flip(-1), x.flip(-2)])
    y = m(xb.to("cuda:0")).cpu()
    assert psnr(y[0], ref[0]) >= PSNR_MIN
    assert psnr(y[1], O.model_forward(sd, xb[1:2])[0]) >= PSNR_MIN
    assert psnr(y[2], O.model_forward(sd, xb[2:3])[0]) >= PSNR_MIN
    yh = m(x.to("cuda:0").half())
    assert yh.dtype == torch.float16 and psnr(yh.float().cpu(), ref) >= 48.0   # fp16 I/O quantisation on top


def test_batch_invariance_and_determinism(hiplib):
    m, _ = make_model(2, 5)
    x = torch.rand(5, 3, 64, 64, generator=torch.Generator().manual_seed(9)).to("cuda:0")
    y = m(x)
    assert torch.equal(y, m(x)), "non-deterministic"
    assert torch.equal(y[3:4], m(x[3:4])), "result depends on the minibatch size"


def test_tiled_render_matches_reference_fixture(hiplib, golden_swin):
    from nunif_amd.nunif.utils.render import tiled_render
    m, sd = make_model(2, 102)
    img = torch.from_numpy(golden_swin["img"])
    ref = torch.from_numpy(golden_swin["render_2x_t64_b4"])    # reference tiled_render output
    out = tiled_render(img, m, tile_size=64, batch_size=4)
    assert out.shape == ref.shape and out.device.type == "cuda"
    assert psnr(out.cpu(), ref) >= PSNR_MIN
    # minibatch size must not matter; the generic (non-fused) path must agree with the fused frame path
    out2 = tiled_render(img, m, tile_size=64, batch_size=1)
    assert torch.equal(out, out2)
    gen = tiled_render(img, _NoFrame(m), tile_size=64, batch_size=4)
    assert torch.equal(gen, out), "fused gather+conv1 frame path != gather_tiles + forward + stitch"


class _NoFrame(torch.nn.Module):
    """Hide render_frame so that tiled_render takes the generic gather -> model -> stitch path."""

    def __init__(self, m):
        super().__init__()
        self.m = m
        for k in ("i2i_scale", "i2i_offset", "i2i_blend_size", "i2i_default_tile_size", "i2i_default_batch_size"):
            setattr(self, k, getattr(m, k))

    def get_device(self):
        return self.m.get_device()

    def find_valid_tile_size(self, t):
        return self.m.find_valid_tile_size(t)

    def forward(self, x):
        return self.m(x)


def test_full_size_1080p_properties(hiplib):
    """BASELINE config 2 (1080p, tile 256) — too slow for the CPU oracle as a whole frame, so check
    size-independent properties: a cropped window of the frame rendered alone must reproduce the same pixels away
    from its border (tiling/stitch invariance up to fp16 noise), flip equivariance of the grid, determinism, and
    oracle parity on one interior 256-tile."""
    from nunif_amd.nunif.utils.render import tiled_render
    m, sd = make_model(2, 102)
    img = synth_image(77, 3, 1080, 1920)
    out = tiled_render(img, m, tile_size=256, batch_size=8)
    assert out.shape == (3, 2160, 3840)
    assert torch.equal(out, tiled_render(img, m, tile_size=256, batch_size=5))
    # one tile through the oracle: tile (1,1) of the grid covers input [228:484) (pad 8, step 236)
    cfg = OS.create_config(1080, 1920, 2, 16, 256, 8)
    xp = torch.nn.functional.pad(img[None], cfg["pad"], mode="replicate")[0]
    t = xp[:, 236:236 + 256, 236:236 + 256][None]
    z = O.model_forward(sd, t)[0]                       # [3,480,480] covers output [472:952)
    inner = out[:, 472 + 8:952 - 8, 472 + 8:952 - 8].cpu()   # exclude the blended ramps
    assert psnr(inner, z[:, 8:-8, 8:-8]) >= PSNR_MIN
    # the last tile (4,8): mostly replicate padding on the right/bottom; its un-blended interior that lies inside
    # the frame is output rows [1888+8 : 2160), cols [3776+8 : 3840)
    t = xp[:, 4 * 236:4 * 236 + 256, 8 * 236:8 * 236 + 256][None]
    z = O.model_forward(sd, t)[0]
    assert psnr(out[:, 1896:2160, 3784:3840].cpu(), z[:, 8:272, 8:64]) >= PSNR_MIN
    # first tile (0,0): top/left replicate padding of 8 px
    z = O.model_forward(sd, xp[:, 0:256, 0:256][None])[0]
    assert psnr(out[:, 0:472, 0:472].cpu(), z[:, 0:472, 0:472]) >= PSNR_MIN
    # a whole different tile size renders through the same code path
    out208 = tiled_render(img, m, tile_size=208, batch_size=8)
    assert out208.shape == out.shape and float(out208.min()) >= 0 and float(out208.max()) <= 1


def test_timed_configuration_batch45_two_streams_equals_small_batches(hiplib):
    """bench.py's timed configuration — 1080p, tile 256, ALL 45 tiles of the frame in one minibatch, two frames in flight on two
    streams (``ConcurrentRenderer``) — must give the bits of the small-minibatch render the oracle-parity tests above use
    (batch 8), for both frames; and the (0,0) / interior tiles of that very output meet the oracle bound."""
    from nunif_amd.nunif.utils.render import tiled_render
    from nunif_amd.parallel import ConcurrentRenderer
    from nunif_amd.waifu2x.models import swin_unet as M
    sd = O.random_state_dict(102, 2)

    def factory():
        mm = M.SwinUNet2x().eval()
        mm.load_state_dict(sd, strict=True)
        return mm

    pool = ConcurrentRenderer(factory, 2, "cuda:0")
    imgs = [synth_image(77 + i, 3, 1080, 1920).to("cuda:0") for i in range(2)]
    base = [tiled_render(im, pool.models[0], tile_size=256, batch_size=8).clone() for im in imgs]
    for rep in range(3):                                            # back to back: the streams overlap from the second round on
        hs = [pool.submit(lambda m, f: tiled_render(f, m, tile_size=256, batch_size=45), im) for im in imgs]
        outs = [pool.result(h).clone() for h in hs]
        torch.cuda.synchronize()
        for o, b in zip(outs, base):
            assert torch.equal(o, b), f"tile batch 45 on two streams != tile batch 8 (round {rep}): {float((o - b).abs().max())}"
    out = outs[0]
    cfg = OS.create_config(1080, 1920, 2, 16, 256, 8)
    xp = torch.nn.functional.pad(imgs[0].cpu()[None], cfg["pad"], mode="replicate")[0]
    z = O.model_forward(sd, xp[:, 2 * 236:2 * 236 + 256, 4 * 236:4 * 236 + 256][None])[0]          # tile (2,4)
    y0, x0 = 2 * 472, 4 * 472
    assert psnr(out[:, y0 + 8:y0 + 472, x0 + 8:x0 + 472].cpu(), z[:, 8:472, 8:472]) >= PSNR_MIN


def test_full_size_4k_4x_properties(hiplib):
    """BASELINE config 3 geometry: swin_unet 4x on a 2160x3840 frame, tile 256 -> 8640x15360 (170 tiles, 1.6 GB of
    fp32 output).  Whole-frame oracle is minutes of CPU, so: grid facts, determinism across tile batch sizes, range,
    and oracle parity on an interior tile and on the bottom-right (mostly padding) tile."""
    from nunif_amd.nunif.utils.render import tiled_render
    m, sd = make_model(4, 104)
    img = synth_image(78, 3, 2160, 3840)
    cfg = OS.create_config(2160, 3840, 4, 32, 256, 16)
    assert (cfg["h_blocks"], cfg["w_blocks"], cfg["input_tile_step"], cfg["output_tile_step"]) == (10, 17, 236, 944)
    out = tiled_render(img, m, tile_size=256, batch_size=34)
    assert out.shape == (3, 8640, 15360) and float(out.min()) >= 0 and float(out.max()) <= 1
    ref_rows = tiled_render(img, m, tile_size=256, batch_size=7)
    assert torch.equal(out[:, 4000:4200], ref_rows[:, 4000:4200]) and torch.equal(out[:, -64:], ref_rows[:, -64:])
    del ref_rows
    xp = torch.nn.functional.pad(img[None], cfg["pad"], mode="replicate")[0]
    t = xp[:, 3 * 236:3 * 236 + 256, 5 * 236:5 * 236 + 256][None]              # tile (3,5)
    z = O.model_forward(sd, t, NAMES[4])[0]                                      # [3,960,960] -> output [3*944 : +960)
    y0, x0 = 3 * 944, 5 * 944
    assert psnr(out[:, y0 + 16:y0 + 944, x0 + 16:x0 + 944].cpu(), z[:, 16:944, 16:944]) >= PSNR_MIN
    t = xp[:, 9 * 236:9 * 236 + 256, 16 * 236:16 * 236 + 256][None]             # last tile (9,16)
    z = O.model_forward(sd, t, NAMES[4])[0]
    y0, x0 = 9 * 944, 16 * 944
    assert psnr(out[:, y0 + 16:8640, x0 + 16:15360].cpu(), z[:, 16:8640 - y0, 16:15360 - x0]) >= PSNR_MIN


def test_load_save_roundtrip_and_errors(hiplib, tmp_path):
    from nunif_amd.nunif.models import load_model, save_model, create_model
    m, sd = make_model(2, 102)
    p = str(tmp_path / "scale2x.pth")
    save_model(m, p)
    m2, meta = load_model(p, device_ids=[0], weights_only=True)
    m2 = m2.eval()
    assert meta["name"] == "waifu2x.swin_unet_2x" and m2.get_device().type == "cuda"
    x = torch.rand(1, 3, 64, 64).to("cuda:0")
    assert torch.equal(m(x), m2(x))
    with pytest.raises(RuntimeError):
        create_model("waifu2x.swin_unet_2x").load_state_dict({"bogus": torch.zeros(1)})
    with pytest.raises(Exception):
        m(torch.rand(1, 3, 100, 100).to("cuda:0"))      # 100 is not a valid tile size
    with pytest.raises(RuntimeError):
        create_model("waifu2x.swin_unet_2x").eval()(torch.rand(1, 3, 64, 64))   # model on CPU: no fallback


def test_swin_unet_8x(hiplib):
    """waifu2x.swin_unet_8x (reference swin_unet.py:303-321): forward only — its registered tile geometry is inconsistent."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from nunif_amd.nunif.models import create_model
    from nunif_amd.waifu2x.models import swin_unet as M  # noqa: F401
    g = np.load(os.path.join(GOLDEN, "swin_unet_8x.npz"))
    m = create_model("waifu2x.swin_unet_8x").eval()
    assert (m.i2i_scale, m.i2i_offset, m.i2i_blend_size) == (4, 64, 32) and not hasattr(m, "render_frame")
    sd = O.random_state_dict(108, 8)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    x = torch.from_numpy(g["x"])
    y = m(x.to("cuda:0")).cpu()
    ref = torch.from_numpy(g["y"]).float()
    assert y.shape == ref.shape == (1, 3, 384, 384)
    assert psnr(y, ref) >= PSNR_MIN, psnr(y, ref)
    assert psnr(y, O.model_forward(sd, x, "waifu2x.swin_unet_8x")) >= PSNR_MIN


def test_swin_unet_4xl(hiplib):
    """waifu2x.swin_unet_4xl (reference swin_unet.py:390-394: base_dim 192, 12 heads of 16 / 32, LayerNormNoBias) and
    SwinUNet2x(layer_norm=True) on the engine's generic block path, against the reference's own outputs."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from nunif_amd.nunif.models import create_model
    from nunif_amd.waifu2x.models import swin_unet as M
    g = np.load(os.path.join(GOLDEN, "swin_unet_4xl.npz"))
    x = torch.from_numpy(g["x"])
    m = create_model("waifu2x.swin_unet_4xl").eval()
    assert (m.i2i_scale, m.i2i_offset, m.i2i_blend_size) == (4, 32, 16)
    sd = O.random_state_dict(204, 4, base_dim=192, layer_norm=True)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    y = m(x.to("cuda:0")).cpu()
    ref = torch.from_numpy(g["y"]).float()
    assert y.shape == ref.shape == (1, 3, 192, 192)
    assert psnr(y, ref) >= PSNR_MIN, psnr(y, ref)
    # whole-frame tiled render through the same handle (2 x 2 tiles of 64) against the oracle's per-tile render
    m2 = M.SwinUNet2x(layer_norm=True).eval()
    sd2 = O.random_state_dict(205, 2, base_dim=96, layer_norm=True)
    m2.load_state_dict(sd2, strict=True)
    m2 = m2.to("cuda:0")
    y2 = m2(x.to("cuda:0")).cpu()
    ref2 = torch.from_numpy(g["y2_ln"]).float()
    assert psnr(y2, ref2) >= PSNR_MIN, psnr(y2, ref2)
    from nunif_amd.nunif.utils.render import tiled_render
    img = x[0, :, :50, :60].contiguous()
    out = tiled_render(img.to("cuda:0"), m, tile_size=64, batch_size=4).cpu()
    assert out.shape == (3, 200, 240) and torch.isfinite(out).all()


def test_tile_row_sharding_matches_whole_render_bit_exact(hiplib):
    """nunif_hip_swin_unet_render_tile_rows / tile_row_band / stitch_rows: two engine handles play two ranks of
    ``parallel.render_rows_sharded`` on one GPU — each renders its tile rows, the upper one hands the overlap band of its last
    tile row to the lower one, each stitches its band of output rows.  Same tiles, same stitch kernel: torch.equal."""
    from nunif_amd.parallel import render_rows_sharded, tile_row_plan
    m, sd = make_model(2, 102)
    m2, _ = make_model(2, 102)
    x = synth_image(31, 3, 300, 200).to("cuda:0")
    ref = m.render_frame(x, tile_size=64, batch_size=5)
    xa, ea = m.row_engine(x, tile_size=64, batch_size=5)
    xb, eb = m2.row_engine(x, tile_size=64, batch_size=3)
    assert torch.equal(render_rows_sharded(xa, ea), ref)                      # world 1 through the tile-row calls
    plan = tile_row_plan(ea.h_blocks, ea.output_tile_step, ea.out_tile_size, ea.y_h, 2)
    (a0, a1, ya0, ya1), (b0, b1, yb0, yb1) = plan
    assert a0 == 0 and a1 == b0 and b1 == ea.h_blocks and a1 > 0 and b1 > b0
    overlap = ea.out_tile_size - ea.output_tile_step
    ea.render_tile_rows(xa, a0, a1)
    eb.render_tile_rows(xb, b0, b1)
    eb.import_band(b0 - 1, eb.output_tile_step, ea.export_band(a1 - 1, ea.output_tile_step, overlap))
    got = torch.cat([ea.stitch_rows(ya0, ya1), eb.stitch_rows(yb0, yb1)], dim=1)
    assert got.shape == ref.shape and torch.equal(got, ref)
