"""The iw3 frame scheduler on the HIP engine (nunif_amd/iw3/frame_pipeline.py): uint8 host frames -> FrameCallbackPool ->
bind_batch_frame_callback (depth stage on one HIP stream, stereo stage on another) -> quantised SBS frames, against what
the REFERENCE's scheduler produced for the same frames (tests/golden/frame_pool.npz), and stage streams on vs off."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

from conftest import psnr

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_cases import (FRAME_CALLBACK_CASES, FRAME_POOL_CASES, FakeSide, FakeWindowDepth, fake_depth_net,  # noqa: E402
                                frame_pool_frames)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame_pool.npz")


def _args(batch_size, **kw):
    base = dict(batch_size=batch_size, tta=False, low_vram=False, disable_amp=True, edge_dilation=0, depth_aa=False,
                rgbd=False, half_rgbd=False, method="grid_sample", mapper="none", divergence=2.0, convergence=0.5,
                synthetic_view="both", pix_fmt="yuv420p", state={"device": torch.device(DEV)})
    base.update(kw)
    return argparse.Namespace(**base)


def _fake_depth():
    from nunif_amd.iw3.base_depth_model import BaseDepthModel

    class FakeDepth(BaseDepthModel):
        def load_model(self, model_type, resolution=None, device=None, **kw):
            return None

        def is_metric(self):
            return False

        def infer(self, x, **kw):                            # the stand-in net is not under test: same values as the fixture
            y = fake_depth_net(x.cpu() if x.ndim == 4 else x[None].cpu()).to(x.device)
            return y if x.ndim == 4 else y[0]

    dm = FakeDepth("fake")
    dm.device = torch.device(DEV)
    return dm


def _u8(x):
    return (x * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().numpy()       # what a decoder hands over


def _run(name, monkeypatch, stage_streams, **kw):
    from nunif_amd.iw3.frame_pipeline import FrameCallbackPool, HostFrame, bind_batch_frame_callback
    monkeypatch.setenv("NUNIF_IW3_STAGE_STREAMS", "1" if stage_streams else "0")
    n, bs, cuts, ema, workers = FRAME_POOL_CASES[name]
    dm = _fake_depth()
    if ema is not None:
        dm.enable_ema(ema[0], buffer_size=ema[1])
    cb, pre = bind_batch_frame_callback(dm, None, set(cuts), _args(bs, **kw))
    pool = FrameCallbackPool(frame_callback=cb, preprocess_callback=pre, batch_size=bs, device=[torch.device(DEV)],
                             max_workers=workers, max_batch_queue=workers + 1, require_pts=True, require_flush=True)
    counts, frames = [], []
    for i, x in enumerate(frame_pool_frames(n)):
        r = pool(HostFrame(_u8(x), i)) or []
        counts.append(len(r))
        frames += r
    r = pool(None)
    counts.append(len(r))
    frames += r
    torch.cuda.synchronize()
    return counts, torch.stack([f.cpu() for f in frames])


@pytest.mark.parametrize("name", sorted(FRAME_POOL_CASES))
def test_scheduler_matches_reference(hiplib, monkeypatch, name):
    g = np.load(GOLDEN)
    n, bs, cuts, ema, workers = FRAME_POOL_CASES[name]
    counts, frames = _run(name, monkeypatch, True)
    ref = torch.from_numpy(g[("ema" if name == "ema_threads" else name) + "_frames"])      # CHW float, reference output
    ref_u8 = (ref * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)         # VU.to_frame quantisation
    assert frames.shape == ref_u8.shape and frames.dtype == torch.uint8
    diff = (frames.int() - ref_u8.int()).abs()
    assert int(diff.max()) <= 1                              # grid-sample warp is fp32 on both sides: rounding ties only
    assert psnr(frames.float() / 255, ref_u8.float() / 255) >= 50.0
    if workers <= 0:
        assert counts == g[name + "_counts"].tolist()        # frames handed back per call == the reference's schedule
    else:
        assert sum(counts) == n


@pytest.mark.parametrize("method", ["grid_sample", "forward_fill"])
def test_stage_streams_do_not_change_the_frames(hiplib, monkeypatch, method):
    """Depth stage and stereo stage on two HIP streams vs everything on the caller's stream: identical bytes (a missing
    event or an early allocator reuse would show up here)."""
    for _ in range(3):
        _, a = _run("ema", monkeypatch, True, method=method)
        _, b = _run("ema", monkeypatch, False, method=method)
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", sorted(FRAME_CALLBACK_CASES))
def test_single_and_windowed_routes(hiplib, name):
    """bind_single_frame_callback / bind_vda_frame_callback on the HIP ops against the reference's own callbacks."""
    from nunif_amd.iw3.frame_pipeline import HostFrame, bind_single_frame_callback, bind_vda_frame_callback

    class Side(FakeSide):
        def flush(self, enable_amp=True):
            le, re = super().flush(enable_amp)
            return le.to(DEV), re.to(DEV)

    g = np.load(GOLDEN)
    n, bs, cuts, ema = FRAME_CALLBACK_CASES[name]
    if name == "single":
        dm = _fake_depth()
        dm.enable_ema(ema[0], buffer_size=ema[1])
        cb = bind_single_frame_callback(dm, Side(), set(cuts), _args(bs))
    else:
        cb = bind_vda_frame_callback(FakeWindowDepth(), Side(), set(cuts), _args(bs))
    counts, frames = [], []
    for i, x in enumerate(frame_pool_frames(n)):
        r = cb(HostFrame(_u8(x), i)) or []
        counts.append(len(r))
        frames += r
    r = cb(None)
    counts.append(len(r))
    frames += r
    assert counts == g[name + "_counts"].tolist()
    got = torch.stack([f.cpu() for f in frames])
    ref = (torch.from_numpy(g[name + "_frames"]) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert got.shape == ref.shape and int((got.int() - ref.int()).abs().max()) <= 1
    assert psnr(got.float() / 255, ref.float() / 255) >= 50.0


def test_sharded_entry_point_on_one_gpu(hiplib, monkeypatch):
    """``stereo_frames_sharded`` (the N-GPU entry, here world = 1) on the HIP ops == the pool on the same frames: the scalar
    replay of the EMA recurrence on the host gives the ranges the device-side scaler gives."""
    from nunif_amd.iw3 import utils as U
    from nunif_amd.iw3.frame_pipeline import stereo_frames_sharded
    name = "ema"
    n, bs, cuts, ema, _ = FRAME_POOL_CASES[name]
    _, pooled = _run(name, monkeypatch, True)
    args = _args(bs)
    dm = _fake_depth()
    dm.enable_ema(ema[0], buffer_size=ema[1])

    def stereo_fn(xs, ds, reset_pts):
        le, ri = U.apply_divergence(ds, xs, args, None)
        return [U.to_frame_tensor(U.postprocess_image(le[i], ri[i], args)) for i in range(xs.shape[0])]

    frames = [U.to_tensor(_u8(x), device=DEV) for x in frame_pool_frames(n)]
    out = stereo_frames_sharded(frames, list(range(n)), set(cuts), dm, stereo_fn, bs)
    got = torch.stack([f.cpu() for f in out])
    assert got.shape == pooled.shape
    diff = (got.int() - pooled.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3
