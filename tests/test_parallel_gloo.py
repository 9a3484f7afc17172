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
