"""Host logic of the iw3 frame scheduler (nunif_amd/iw3/frame_pipeline.py) on CPU.

The fixture ``tests/golden/frame_pool.npz`` holds what the REFERENCE's ``VU.FrameCallbackPool`` +
``iw3.utils.bind_batch_frame_callback`` hand back, call by call, around a fake depth net and the real grid-sample warp
(``tests/golden/make_golden.py::gen_frame_pool``).  Here the same frames go through the product scheduler with torch-CPU
stand-ins for the device functions (the HIP ops refuse CPU tensors): the schedule (frames per call) must be identical and
the frames equal.  The N > 1 path (``stereo_frames_sharded``) runs on 2 gloo ranks and must be bit-identical to the single
process.
"""
import argparse
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_cases import (FRAME_CALLBACK_CASES, FRAME_POOL_CASES, FakeSide, FakeWindowDepth, fake_depth_net,  # noqa: E402
                                frame_pool_frames)

from nunif_amd.iw3.base_depth_model import BaseDepthModel  # noqa: E402
from nunif_amd.iw3.frame_pipeline import (FrameCallbackPool, PipelineOps, bind_batch_frame_callback,  # noqa: E402
                                          bind_single_frame_callback, bind_vda_frame_callback, stereo_frames_sharded)
from oracle.backward_warp import grid_sample_warp  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame_pool.npz")


class Frame:
    def __init__(self, x, pts):
        self.x, self.pts = x, pts


class FakeDepth(BaseDepthModel):
    def load_model(self, model_type, resolution=None, device=None, **kw):
        return None

    def is_metric(self):
        return False

    def infer(self, x, **kw):
        return fake_depth_net(x) if x.ndim == 4 else fake_depth_net(x[None])[0]


def _args(batch_size):
    return argparse.Namespace(batch_size=batch_size, tta=False, low_vram=False, disable_amp=True, edge_dilation=0,
                              depth_aa=False, rgbd=False, half_rgbd=False, method="grid_sample", mapper="none",
                              divergence=2.0, convergence=0.5, synthetic_view="both", pix_fmt="yuv420p",
                              state={"device": torch.device("cpu")})


def _cpu_ops():
    """torch-CPU stand-ins with the reference's semantics for this configuration (mapper "none", grid_sample, full SBS)."""
    def apply_divergence(depths, x, args, side_model=None, reset_pts=None):
        if depths.ndim == 3:
            le, re = grid_sample_warp(x[None], depths[None], args.divergence, args.convergence, args.synthetic_view)
            return le[0], re[0]
        return grid_sample_warp(x, depths, args.divergence, args.convergence, args.synthetic_view)

    return PipelineOps(to_tensor=lambda frame, device=None: frame.x,
                       preprocess_image=lambda x, args: x,
                       apply_divergence=apply_divergence,
                       postprocess_image=lambda le, re, args: torch.clamp(torch.cat([le, re], dim=2), 0, 1),
                       to_frame=lambda x, use_16bit=False: x)


def _model(ema):
    dm = FakeDepth("fake")
    if ema is not None:
        dm.enable_ema(ema[0], buffer_size=ema[1])
    return dm


def _run_pool(n, bs, cuts, ema, workers):
    ops = _cpu_ops()
    cb, pre = bind_batch_frame_callback(_model(ema), None, set(cuts), _args(bs), ops=ops)
    pool = FrameCallbackPool(frame_callback=cb, preprocess_callback=pre, batch_size=bs, device=[torch.device("cpu")],
                             max_workers=workers, max_batch_queue=workers + 1, require_pts=True, require_flush=True, ops=ops)
    counts, frames = [], []
    for i, x in enumerate(frame_pool_frames(n)):
        r = pool(Frame(x, i)) or []
        counts.append(len(r))
        frames += r
    r = pool(None)
    counts.append(len(r))
    frames += r
    pool.shutdown()
    return counts, frames


@pytest.mark.parametrize("name", sorted(FRAME_POOL_CASES))
def test_pool_matches_reference_scheduler(name):
    g = np.load(GOLDEN)
    n, bs, cuts, ema, workers = FRAME_POOL_CASES[name]
    counts, frames = _run_pool(n, bs, cuts, ema, workers)
    assert len(frames) == n
    ref = torch.from_numpy(g[("ema" if name == "ema_threads" else name) + "_frames"])
    assert torch.equal(torch.stack(frames), ref)          # same torch ops in the same order: bit-identical
    if workers <= 0:                                       # the synchronous schedule is deterministic: frames per call
        assert counts == g[name + "_counts"].tolist()
    else:                                                  # (on CPU a batch is complete when its call returns)
        assert sum(counts) == n


def test_pool_round_robins_batches_over_several_devices_and_skips_pts():
    ops = _cpu_ops()
    with pytest.raises(ValueError):
        FrameCallbackPool(lambda *a: [], 2, device=[], ops=ops)
    # the reference's ``--gpu 0 1``: whole batches go to the listed devices in turn, frames come back in order
    # (a fake device table: to_tensor records which device each frame was sent to)
    sent = []
    ops2 = _cpu_ops()
    ops2.to_tensor = lambda frame, device=None: (sent.append(str(device)) or frame.x)
    pool2 = FrameCallbackPool(lambda batch, pts: [b for b in batch], 2, device=["cpu", "meta"], max_workers=0,
                              require_pts=True, ops=ops2)
    got = []
    xs = frame_pool_frames(7)
    for i, x in enumerate(xs):
        got += pool2(Frame(x, i)) or []
    got += pool2(None)
    assert sent == ["cpu", "cpu", "meta", "meta", "cpu", "cpu", "meta"]
    assert len(got) == 7 and all(torch.equal(a, b) for a, b in zip(got, xs))
    seen = []
    pool = FrameCallbackPool(lambda batch, pts: (seen.append(list(pts)) or [b for b in batch]), 2, device="cpu",
                             max_workers=0, require_pts=True, skip_pts=1, ops=ops)
    out = []
    for i, x in enumerate(frame_pool_frames(5)):
        out += pool(Frame(x, i)) or []
    out += pool(None)
    assert seen == [[2, 3], [4]] and len(out) == 3         # pts 0 and 1 skipped; the partial batch is flushed at the end


@pytest.mark.parametrize("name", sorted(FRAME_CALLBACK_CASES))
def test_single_and_windowed_routes_match_reference(name):
    """bind_single_frame_callback (per-frame + EMA look-ahead + side-model flush at a cut) and bind_vda_frame_callback
    (a depth model that lags and normalises by itself) against the reference's own callbacks, call by call."""
    g = np.load(GOLDEN)
    n, bs, cuts, ema = FRAME_CALLBACK_CASES[name]
    if name == "single":
        cb = bind_single_frame_callback(_model(ema), FakeSide(), set(cuts), _args(bs), ops=_cpu_ops())
    else:
        cb = bind_vda_frame_callback(FakeWindowDepth(), FakeSide(), set(cuts), _args(bs), ops=_cpu_ops())
    counts, frames = [], []
    for i, x in enumerate(frame_pool_frames(n)):
        r = cb(Frame(x, i)) or []
        counts.append(len(r))
        frames += r
    r = cb(None)
    counts.append(len(r))
    frames += r
    assert counts == g[name + "_counts"].tolist()
    # (the reference feeds its depth net a permuted HWC view on these routes, so the fake net's mean() sums in another order:
    #  last-bit differences in the depth, nothing to do with the scheduler)
    assert (torch.stack(frames) - torch.from_numpy(g[name + "_frames"])).abs().max().item() < 1e-5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stereo_fn(args):
    def fn(xs, ds, reset_pts):
        le, re = grid_sample_warp(xs, ds, args.divergence, args.convergence, "both")
        return [torch.clamp(torch.cat([le[i], re[i]], dim=2), 0, 1) for i in range(xs.shape[0])]
    return fn


def _worker(rank, world, port, case, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, bs, cuts, ema, _ = FRAME_POOL_CASES[case]
    frames = frame_pool_frames(n)
    mine = {i for b in range(rank, (n + bs - 1) // bs, world) for i in range(b * bs, min(n, (b + 1) * bs))}
    frames = [f if i in mine else None for i, f in enumerate(frames)]      # a rank only holds the frames it owns
    with torch.inference_mode():
        out = stereo_frames_sharded(frames, list(range(n)), set(cuts), _model(ema), _stereo_fn(_args(bs)), bs, dst=0)
    if rank == 0:
        torch.save(torch.stack(out), path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["plain", "ema", "cut_last_of_batch"])
def test_sharded_two_ranks_bit_identical_to_sequential(name, tmp_path):
    g = np.load(GOLDEN)
    path = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(2, _free_port(), name, path), nprocs=2, join=True)
    got = torch.load(path)
    assert torch.equal(got, torch.from_numpy(g[name + "_frames"]))


@pytest.mark.parametrize("name", ["plain", "ema", "cut_last_of_batch"])
def test_sharded_single_process(name):
    g = np.load(GOLDEN)
    n, bs, cuts, ema, _ = FRAME_POOL_CASES[name]
    out = stereo_frames_sharded(frame_pool_frames(n), list(range(n)), set(cuts), _model(ema), _stereo_fn(_args(bs)), bs)
    assert torch.equal(torch.stack(out), torch.from_numpy(g[name + "_frames"]))


def _stereo_fn_16bit(args):
    """Finished frames as the engine's ``to_frame(use_16bit=True)`` makes them for 10 / 12 / 16-bit pix_fmts: HWC ``int16`` holding
    the uint16 bit patterns (nunif_amd/iw3/_ops.py; reference ``VU.to_frame`` nunif/utils/video.py:236-245)."""
    base = _stereo_fn(args)

    def fn(xs, ds, reset_pts):
        return [(f.permute(1, 2, 0) * 65535.0).round().to(torch.int32).to(torch.uint16).view(torch.int16) for f in base(xs, ds, reset_pts)]
    return fn


def _worker_16bit(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, bs, cuts, ema, _ = FRAME_POOL_CASES["ema"]
    frames = frame_pool_frames(n)
    mine = {i for b in range(rank, (n + bs - 1) // bs, world) for i in range(b * bs, min(n, (b + 1) * bs))}
    frames = [f if i in mine else None for i, f in enumerate(frames)]
    with torch.inference_mode():
        out = stereo_frames_sharded(frames, list(range(n)), set(cuts), _model(ema), _stereo_fn_16bit(_args(bs)), bs, dst=0)
    if rank == 0:
        torch.save(torch.stack(out), path)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_two_ranks_move_16bit_frames(tmp_path):
    """``iw3 --gpu 0 1 --pix-fmt yuv420p10le``: the finished frames are int16 and neither gloo nor nccl gathers Short tensors — they
    travel as bytes.  Two ranks == one process, bit for bit, dtype kept."""
    path = str(tmp_path / "out.pt")
    mp.spawn(_worker_16bit, args=(2, _free_port(), path), nprocs=2, join=True)
    n, bs, cuts, ema, _ = FRAME_POOL_CASES["ema"]
    ref = stereo_frames_sharded(frame_pool_frames(n), list(range(n)), set(cuts), _model(ema), _stereo_fn_16bit(_args(bs)), bs)
    got = torch.load(path)
    assert got.dtype == torch.int16 and torch.equal(got, torch.stack(ref))


def test_shard_lag_is_bounded(monkeypatch):
    from nunif_amd.iw3.frame_pipeline import ShardedStereoStream
    for raw, want in (("0", 0), ("2", 2), ("4", 4), ("7", 1), ("-3", 1), ("x", 1)):
        monkeypatch.setenv("NUNIF_SHARD_LAG", raw)
        assert ShardedStereoStream(_model(None), lambda *a: [], 2, set())._lag == want


def _worker_idle(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = frame_pool_frames(2)
    frames = frames if rank == 0 else [None, None]                     # one batch only: rank 1 owns nothing
    with torch.inference_mode():
        out = stereo_frames_sharded(frames, [0, 1], set(), _model((0.5, 2)), _stereo_fn(_args(2)), 2, dst=0)
    if rank == 0:
        torch.save(torch.stack(out), path)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rank_without_frames_still_joins(tmp_path):
    path = str(tmp_path / "out.pt")
    mp.spawn(_worker_idle, args=(2, _free_port(), path), nprocs=2, join=True)
    ref = stereo_frames_sharded(frame_pool_frames(2), [0, 1], set(), _model((0.5, 2)), _stereo_fn(_args(2)), 2)
    assert torch.equal(torch.load(path), torch.stack(ref))


def test_sharded_refuses_temporal_models():
    dm = _model(None)
    dm.has_temporal_state = True
    with pytest.raises(ValueError):
        stereo_frames_sharded(frame_pool_frames(2), [0, 1], set(), dm, lambda *a: [], 2)


class _QueueSide:
    """A side model with a temporal queue (the shape of ForwardInpaintVideo / MLBWInpaintVideo): it swallows frames until
    ``depth`` of them are waiting, answers with all of them at once, and hands out the rest on ``flush``."""

    def __init__(self, depth=5):
        self.depth, self.q = depth, []

    def push(self, le, re):
        self.q += list(zip(le, re))
        if len(self.q) < self.depth:
            return None, None
        out, self.q = self.q, []
        return torch.stack([a for a, _ in out]), torch.stack([b for _, b in out])

    def flush(self, enable_amp=True):
        if not self.q:
            return None, None
        out, self.q = self.q, []
        return torch.stack([a for a, _ in out]), torch.stack([b for _, b in out])


def test_batch_route_flushes_a_queueing_side_model():
    """frames in == frames out when the side model keeps a temporal queue: the end-of-stream flush of the batch route has
    to drain it (the reference only sends such models through the single-frame route, iw3/utils.py:658-663)."""
    side = _QueueSide(depth=5)
    ops = _cpu_ops()
    base = ops.apply_divergence

    def apply_divergence(depths, x, args, side_model=None, reset_pts=None):
        le, re = base(depths, x, args, None, reset_pts)
        return side_model.push(le, re)

    ops.apply_divergence = apply_divergence
    n, bs = 13, 3
    cb, pre = bind_batch_frame_callback(_model(None), side, set(), _args(bs), ops=ops)
    pool = FrameCallbackPool(frame_callback=cb, preprocess_callback=pre, batch_size=bs, device=[torch.device("cpu")],
                             max_workers=0, max_batch_queue=1, require_pts=True, require_flush=True, ops=ops)
    frames = []
    xs = frame_pool_frames(n)
    for i, x in enumerate(xs):
        frames += pool(Frame(x, i)) or []
    frames += pool(None)
    pool.shutdown()
    assert len(frames) == n and not side.q
    # order and content: the same frames as the queue-less route
    _, ref = _run_pool(n, bs, (), None, 0)
    assert torch.equal(torch.stack(frames), torch.stack(ref))


def test_engine_models_deepcopy_without_their_handle():
    """Model.to_inference_model() deep-copies; an engine-backed model holds a ctypes handle once it has run."""
    import copy
    import ctypes
    from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x, SwinUNet4x
    m = SwinUNet2x()
    m._engine = ctypes.c_void_p(1234)                    # what a built engine looks like to copy.deepcopy
    c = m.to_inference_model()
    assert c._engine is None and c is not m and not c.training
    a, b = m.state_dict(), c.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    k0 = next(iter(a))
    c._weights[k0] += 1.0
    assert not torch.equal(m._weights[k0], c._weights[k0])          # own storage
    m._engine = None
    m4 = SwinUNet4x()
    assert m4.to_2x(shared=True).net4x is m4 and m4.to_2x(shared=False).net4x is not m4
    assert copy.deepcopy(m4.to_1x(shared=False)).net4x is not m4
