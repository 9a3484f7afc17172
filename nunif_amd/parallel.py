"""Frame-sharded multi-GPU execution: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

The reference has no collective backend: it wraps the tile minibatch in ``nn.DataParallel``
(``nunif/models/data_parallel.py:41-50``) or round-robins frame batches over per-device replicas from a thread pool
(``nunif/utils/video.py:1622-1757``, ``iw3/utils.py:709-831``).  Frames are independent, so here each
rank renders frames ``rank, rank+world, ...`` entirely on its own GPU — tiles, stitch and quantisation included —
and the only communication is the ordered gather of finished, already-quantised frames to the I/O rank
(uint8/uint16 HWC, exactly what ``VU.to_frame`` hands to the encoder, ``nunif/utils/video.py:236-245``).  There is no
all-reduce anywhere.
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Indices of the items rank ``rank`` owns (round-robin keeps every rank busy for any n)."""
    return list(range(rank, n_items, world_size))


def to_frame(x, bits=8):
    """CHW float [0,1] -> HWC uint8/uint16, round-to-nearest like ``VU.to_frame`` (video.py:236-245)."""
    maxv = 255.0 if bits == 8 else 65535.0
    q = torch.clamp(torch.round(x.float() * maxv), 0, maxv)
    if bits == 8:
        return q.to(torch.uint8).permute(1, 2, 0).contiguous()
    return q.to(torch.int32).permute(1, 2, 0).contiguous()        # torch has no uint16 arithmetic; widen


def render_sharded(frames, render_fn, group=None, dst=0, bits=8):
    """Render ``frames`` (sequence of CHW tensors, identical on every rank) frame-sharded across the group.

    ``render_fn(frame) -> CHW float tensor in [0,1]`` runs on the calling rank's device.  Returns, on ``dst``, the
    list of quantised HWC frames in the original order; ``None`` on the other ranks.  Works for any backend
    (``nccl`` == RCCL on ROCm; ``gloo`` in the CPU tests)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [to_frame(render_fn(f), bits) for f in frames]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_indices(len(frames), rank, world)
    local = [to_frame(render_fn(frames[i]), bits) for i in mine]
    per_rank = (len(frames) + world - 1) // world
    # every rank contributes a fixed-size [per_rank, H, W, C] block (padding with zeros) -> one gather
    if local:
        shape, dtype, device = local[0].shape, local[0].dtype, local[0].device
    else:
        probe = to_frame(render_fn(frames[0]), bits)                # shape discovery for an idle rank
        shape, dtype, device = probe.shape, probe.dtype, probe.device
    block = torch.zeros((per_rank, *shape), dtype=dtype, device=device)
    for k, f in enumerate(local):
        block[k] = f
    if rank == dst:
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.gather(block, parts, dst=dst, group=group)
        out = [None] * len(frames)
        for r in range(world):
            for k, i in enumerate(shard_indices(len(frames), r, world)):
                out[i] = parts[r][k]
        return out
    dist.gather(block, None, dst=dst, group=group)
    return None
