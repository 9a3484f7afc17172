"""Frame sharding + ordered gather on 2 CPU ranks over gloo (the N>1 path of bench.py / nunif_amd.parallel)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nunif_amd.parallel import render_sharded, shard_indices, to_frame


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_render(frame):
    up = torch.nn.functional.interpolate(frame[None], scale_factor=2, mode="nearest")[0]
    return up * 0.5 + 0.25


def _worker(rank, world, port, n_frames, result_path, streaming=False, dst=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    frames = [torch.rand(3, 10, 12, generator=g) for _ in range(n_frames)]
    if streaming:
        got = {}
        out = render_sharded(frames, _fake_render, dst=dst, on_frame=lambda i, f: got.__setitem__(i, f.clone()))
        assert out is None
        if rank == dst:
            out = [got[i] for i in range(n_frames)]
    else:
        out = render_sharded(frames, _fake_render, dst=dst)
    if rank == dst:
        torch.save(out, result_path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 5, 8, 17):
        for w in (1, 2, 3, 8):
            seen = sorted(i for r in range(w) for i in shard_indices(n, r, w))
            assert seen == list(range(n))


def test_to_frame_rounding():
    x = torch.tensor([[[0.0, 0.5, 1.0, 1.2, -0.1, 0.49803921]]]).expand(3, 1, 6)
    assert to_frame(x)[0, :, 0].tolist() == [0, 128, 255, 255, 0, 127]
    assert to_frame(x, bits=16)[0, 2, 0].item() == 65535


def test_render_sharded_two_ranks_matches_single_process(tmp_path):
    for n_frames in (5, 1):        # 1 frame: rank 1 is idle and still has to join the gather
        path = str(tmp_path / f"out{n_frames}.pt")
        mp.spawn(_worker, args=(2, _free_port(), n_frames, path), nprocs=2, join=True)
        got = torch.load(path)
        g = torch.Generator().manual_seed(7)
        frames = [torch.rand(3, 10, 12, generator=g) for _ in range(n_frames)]
        ref = [to_frame(_fake_render(f)) for f in frames]
        assert len(got) == n_frames
        for a, b in zip(got, ref):
            assert a.dtype == torch.uint8 and torch.equal(a, b)


def test_render_sharded_three_ranks_streaming_and_nonzero_dst(tmp_path):
    """world 3, 7 frames (uneven shards), the streaming ``on_frame`` form, and an I/O rank other than 0."""
    for streaming, dst in ((True, 0), (False, 2)):
        path = str(tmp_path / f"out_{int(streaming)}_{dst}.pt")
        mp.spawn(_worker, args=(3, _free_port(), 7, path, streaming, dst), nprocs=3, join=True)
        got = torch.load(path)
        g = torch.Generator().manual_seed(7)
        ref = [to_frame(_fake_render(torch.rand(3, 10, 12, generator=g))) for _ in range(7)]
        assert len(got) == 7
        for a, b in zip(got, ref):
            assert torch.equal(a, b)


# ---- tile-row sharding of ONE image (render_rows_sharded) --------------------------------------------------------------------
class _FakeRowEngine:
    """A torch-CPU stand-in with the SwinRowEngine interface: tile outputs are a deterministic function of the tile's pixels,
    the stitch is the reference's running-mean recurrence (oracle.seam_blending) over whatever the tile store holds."""

    def __init__(self, H, W, scale=2, offset=16, blend=8, tile=64):
        from oracle import seam_blending as OS
        self.OS, self.scale, self.offset, self.blend, self.tile = OS, scale, offset, blend, tile
        cfg = OS.create_config(H, W, scale, offset, tile, blend)
        self.cfg = cfg
        self.h_blocks, self.w_blocks = cfg["h_blocks"], cfg["w_blocks"]
        self.out_tile_size, self.output_tile_step = tile * scale - 2 * offset, cfg["output_tile_step"]
        self.y_h, self.y_w = cfg["y_h"], cfg["y_w"]
        self.device = torch.device("cpu")
        self.store = torch.full((self.h_blocks * self.w_blocks, 3, self.out_tile_size, self.out_tile_size), float("nan"))

    def _net(self, mb):
        z = torch.nn.functional.interpolate(mb, scale_factor=self.scale, mode="nearest")
        z = z * 0.6 + 0.3 * mb.mean(dim=(1, 2, 3), keepdim=True)
        o = self.offset
        return z[:, :, o:-o, o:-o]

    def render_tile_rows(self, x, r0, r1):
        xp = torch.nn.functional.pad(x[None], self.cfg["pad"], mode="replicate")[0]
        st, T = self.cfg["input_tile_step"], self.tile
        for i in range(r0, r1):
            for j in range(self.w_blocks):
                self.store[i * self.w_blocks + j] = self._net(xp[None, :, i * st:i * st + T, j * st:j * st + T])[0]

    def export_band(self, tile_row, row0, n):
        return self.store[tile_row * self.w_blocks:(tile_row + 1) * self.w_blocks, :, row0:row0 + n].clone()

    def import_band(self, tile_row, row0, band):
        self.store[tile_row * self.w_blocks:(tile_row + 1) * self.w_blocks, :, row0:row0 + band.shape[2]] = band

    def stitch_rows(self, y0, y1):
        OS, To, ostep = self.OS, self.out_tile_size, self.output_tile_step
        px = torch.zeros(3, self.cfg["y_buffer_h"], self.cfg["y_buffer_w"])
        wt = torch.zeros_like(px)
        filt = OS.blend_filter(self.scale, self.offset, self.tile, self.blend, 3)
        for i in range(self.h_blocks):
            if i * ostep >= y1 or i * ostep + To <= y0:
                continue                                       # this tile row does not touch the band
            for j in range(self.w_blocks):
                t = self.store[i * self.w_blocks + j]
                ys, xs = slice(ostep * i, ostep * i + To), slice(ostep * j, ostep * j + To)
                w_old = wt[:, ys, xs]
                w_new = w_old + filt
                a = w_old / w_new
                px[:, ys, xs] = px[:, ys, xs] * a + t * (1 - a)
                wt[:, ys, xs] = w_new
        return torch.clamp(px[:, y0:y1, :self.y_w], 0, 1).contiguous()


def _rows_worker(rank, world, port, H, W, path, dst):
    from nunif_amd.parallel import render_rows_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.rand(3, H, W, generator=torch.Generator().manual_seed(11))
    out = render_rows_sharded(x, _FakeRowEngine(H, W), dst=dst)
    if rank == dst:
        torch.save(out, path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_tile_row_plan_covers_rows_and_tiles_once():
    from nunif_amd.parallel import tile_row_plan
    for hb, ostep, To, y_h in ((5, 472, 480, 2160), (10, 944, 960, 8640), (1, 96, 100, 90), (3, 96, 96, 250)):
        for world in (1, 2, 3, 8):
            plan = tile_row_plan(hb, ostep, To, y_h, world)
            assert len(plan) == world and plan[0][0] == 0 and plan[-1][1] == hb
            assert all(a[1] == b[0] and a[3] == b[2] for a, b in zip(plan, plan[1:]))
            assert plan[0][2] == 0 and plan[-1][3] == y_h
            assert all(r0 <= r1 and y0 <= y1 for r0, r1, y0, y1 in plan)


def test_render_rows_sharded_is_bit_identical_to_the_whole_render(tmp_path):
    """2 and 3 ranks (and more ranks than tile rows): the band exchange + per-rank stitch reproduces the whole-frame render bit
    for bit, because tiles, recurrence order and arithmetic are the same."""
    from nunif_amd.parallel import render_rows_sharded
    for H, W, world, dst in ((150, 100, 2, 0), (230, 70, 3, 1), (60, 90, 3, 0)):
        x = torch.rand(3, H, W, generator=torch.Generator().manual_seed(11))
        ref = render_rows_sharded(x, _FakeRowEngine(H, W))                 # world 1: render everything, stitch everything
        path = str(tmp_path / f"rows_{H}_{world}.pt")
        mp.spawn(_rows_worker, args=(world, _free_port(), H, W, path, dst), nprocs=world, join=True)
        got = torch.load(path)
        assert got.shape == ref.shape == (3, 2 * H, 2 * W)
        assert torch.equal(got, ref), (H, W, world)


def _nccl_worker(rank, world, port, n_frames, result_path, dst):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    g = torch.Generator().manual_seed(7)
    frames = [torch.rand(3, 10, 12, generator=g).to(f"cuda:{rank}") for _ in range(n_frames)]
    out = render_sharded(frames, _fake_render, dst=dst)
    if rank == dst:
        torch.save([o.cpu() for o in out], result_path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_render_sharded_two_gpus_nccl(tmp_path):
    """The same delivery path under RCCL (lazy per-pair communicators, grouped receives against single sends, buffer lifetime
    after ``wait``): needs two GPUs, skipped on the one-GPU test box; n not divisible by the world and a dst that owns 0 frames."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    for n_frames, dst in ((5, 0), (1, 1)):
        path = str(tmp_path / f"nccl_{n_frames}_{dst}.pt")
        mp.spawn(_nccl_worker, args=(2, _free_port(), n_frames, path, dst), nprocs=2, join=True)
        got = torch.load(path)
        g = torch.Generator().manual_seed(7)
        ref = [to_frame(_fake_render(torch.rand(3, 10, 12, generator=g))) for _ in range(n_frames)]
        assert len(got) == n_frames and all(torch.equal(a, b) for a, b in zip(got, ref))


def _evidence_worker(rank, world, port, path):
    import importlib.util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ev = bench.collect_multi_gpu(dist, world, rank, rank, torch.device("cpu"))
    if rank == 0:
        torch.save(ev, path)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multi_gpu_evidence_block_on_two_gloo_ranks(tmp_path):
    """bench.py --gpus N prints which ranks / devices took part (the driver's SCALE runs are the only N > 1 executions this code
    ever gets): the collecting function on two CPU ranks."""
    path = str(tmp_path / "ev.pt")
    mp.spawn(_evidence_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    ev = torch.load(path)
    assert ev["ranks_seen"] == 2 and ev["world_size"] == 2 and ev["distinct_pci_bus_ids"] == 2 and ev["backend"] == "gloo"
    assert [d["rank"] for d in ev["devices"]] == [0, 1]


def _iw3_leg_worker(rank, world, port, path):
    import argparse
    import importlib.util
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_golden_cases import fake_depth_net, frame_pool_frames
    from nunif_amd.iw3.base_depth_model import BaseDepthModel
    from oracle.backward_warp import grid_sample_warp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(here), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakeDepth(BaseDepthModel):
        def load_model(self, model_type, resolution=None, device=None, **kw):
            return None

        def is_metric(self):
            return False

        def infer(self, x, **kw):
            return fake_depth_net(x)

    def stereo_fn(xs, ds, reset_pts):
        le, re = grid_sample_warp(xs, ds, 2.0, 0.5, "both")
        return [(torch.clamp(torch.cat([le[i], re[i]], dim=2), 0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0) for i in range(xs.shape[0])]

    pool = frame_pool_frames(4)
    rec = bench.iw3_sharded_leg(dist, world, rank, torch.device("cpu"), dist.barrier, frames_per_rank=8, batch=2,
                                depth_model=FakeDepth("fake"), stereo_fn=stereo_fn, frame_hw=(24, 40), make_frame=lambda i: pool[i % 4])
    if rank == 0:
        torch.save(rec, path)
    else:
        assert rec is None
    dist.barrier()
    dist.destroy_process_group()


def test_bench_iw3_leg_at_two_ranks_prints_a_record(tmp_path):
    """``bench.py --gpus N`` carries BASELINE's iw3 metric at N > 1: the leg on two gloo ranks with CPU
    stand-ins for the depth net and the warp — every frame of the one stream arrives on rank 0, the record names the sharding."""
    path = str(tmp_path / "rec.pt")
    mp.spawn(_iw3_leg_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    rec = torch.load(path)
    assert rec["world"] == 2 and rec["frames"] == 16 and rec["frames_delivered"] == 16 and rec["value"] > 0
    assert "stereo_frames_sharded" in rec["config"] and rec["unit"] == "input MPix/s"


def _config5_leg_worker(rank, world, port, path):
    import importlib.util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(here), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def fake_record(dev):           # what config5_record returns, rank 1 the slower stream
        ms = 20.0 + 5.0 * rank
        return {"config": "fake config 5", "depth_net": "vda_streaming_vits", "frame": [2160, 3840], "frames_in": 24, "frames_out": 24,
                "ms_per_frame": ms, "fps": round(1e3 / ms, 1), "value": 1.0, "unit": "input MPix/s", "target": "t"}

    rec = bench.config5_replicas_leg(dist, world, rank, torch.device("cpu"), dist.barrier, record_fn=fake_record)
    if rank == 0:
        torch.save(rec, path)
    else:
        assert rec is None
    dist.barrier()
    dist.destroy_process_group()


def test_bench_config5_leg_runs_replicas_and_prices_them_at_the_slowest_rank(tmp_path):
    """``bench.py --gpus N``: config 5 does not shard (temporal depth net, 12-frame inpaint queue; the reference runs it on one GPU) —
    N independent streams, ``value`` at the slowest rank's time per frame."""
    path = str(tmp_path / "rec.pt")
    mp.spawn(_config5_leg_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    rec = torch.load(path)
    assert rec["world"] == 2 and rec["ms_per_frame_per_gpu"] == 25.0 and rec["fps"] == 80.0
    assert rec["value"] == round(2160 * 3840 * 2 / 25.0 / 1e3, 1) and rec["scaling"].startswith("replicas")
    assert [r["rank"] for r in rec["per_rank"]] == [0, 1] and rec["depth_net"] == "vda_streaming_vits"
