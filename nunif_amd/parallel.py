"""Frame-sharded multi-GPU execution: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

The reference has no collective backend: it wraps the tile minibatch in ``nn.DataParallel``
(``nunif/models/data_parallel.py:41-50``) or round-robins frame batches over per-device replicas from a thread pool
(``nunif/utils/video.py:1622-1757``, ``iw3/utils.py:709-831``).  Frames are independent, so here each
rank renders frames ``rank, rank+world, ...`` entirely on its own GPU — tiles, stitch and quantisation included —
and the only communication is the delivery of finished, already-quantised frames to the I/O rank (uint8/uint16 HWC,
exactly what ``VU.to_frame`` hands to the encoder, ``nunif/utils/video.py:236-245``): one non-blocking point-to-point
send per frame, overlapped with the next render.  There is no all-reduce anywhere.
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Indices of the items rank ``rank`` owns (round-robin keeps every rank busy for any n)."""
    return list(range(rank, n_items, world_size))


def to_frame(x, bits=8):
    """CHW float [0,1] -> HWC uint8/uint16, round-to-nearest like ``VU.to_frame`` (video.py:236-245).

    Device tensors go through the HIP quantise kernel (``nunif_hip_stereo_to_frame``, one pass: clamp, scale, round,
    CHW -> HWC); host tensors (the gloo tests, the oracle side of a comparison) through the same arithmetic in torch."""
    if x.device.type == "cuda":
        from .iw3 import _ops
        q = _ops.to_frame(x.float().contiguous(), bits)
        return q if bits == 8 else q.to(torch.int32) & 0xFFFF           # uint16 bit pattern -> widened like the host path
    maxv = 255.0 if bits == 8 else 65535.0
    q = torch.clamp(torch.round(x.float() * maxv), 0, maxv)
    if bits == 8:
        return q.to(torch.uint8).permute(1, 2, 0).contiguous()
    return q.to(torch.int32).permute(1, 2, 0).contiguous()        # torch has no uint16 arithmetic; widen


def render_sharded(frames, render_fn, group=None, dst=0, bits=8, on_frame=None):
    """Render ``frames`` (sequence of CHW tensors, identical on every rank) frame-sharded across the group.

    ``render_fn(frame) -> CHW float tensor in [0,1]`` runs on the calling rank's device.  Rank r owns frames
    r, r + world, ...; every finished frame is quantised on the device and SENT to ``dst`` right away with a
    non-blocking point-to-point transfer (RCCL over xGMI under the ``nccl`` backend), so the transfer of frame k overlaps
    the render of frame k + 1 and nothing but the frames in flight is held on the workers — no padded block, no
    end-of-job collective.  ``dst`` posts the matching receives round by round (one grouped launch per round) into
    buffers of exactly the frame size.  Returns, on ``dst``, the list of quantised HWC frames in the original order
    (``on_frame(i, frame)`` is called instead of collecting when given — the streaming form); ``None`` elsewhere.
    Works for any backend (``gloo`` in the CPU tests)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        out = []
        for i, f in enumerate(frames):
            q = to_frame(render_fn(f), bits)
            if on_frame is not None:
                on_frame(i, q)
            else:
                out.append(q)
        return None if on_frame is not None else out
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = len(frames)
    # dst sizes its receive buffers from ITS OWN frames: unequal frame sizes would corrupt silently, so refuse them up front
    shapes = {tuple(f.shape) for f in frames if f is not None}
    if len(shapes) > 1:
        raise ValueError(f"render_sharded: all frames must have one shape, got {sorted(shapes)}")
    rounds = (n + world - 1) // world
    mine = shard_indices(n, rank, world)
    out = [None] * n
    pending = []                      # (requests, [(index, buffer)]) of earlier rounds

    def deliver(done):
        for i, buf in done:
            if on_frame is not None:
                on_frame(i, buf)
            else:
                out[i] = buf

    shape = dtype = device = None
    for k in range(rounds):
        q = None
        if k < len(mine):
            q = to_frame(render_fn(frames[mine[k]]), bits)
            shape, dtype, device = q.shape, q.dtype, q.device
        if rank != dst:
            if q is not None:
                pending.append((dist.batch_isend_irecv([dist.P2POp(dist.isend, q, dst, group)]), q))
            # at most two sends in flight: the buffer of round k - 2 is released before round k renders
            while len(pending) > 2:
                reqs, _ = pending.pop(0)
                for r in reqs:
                    r.wait()
            continue
        # ---- dst: receives of this round (peers whose k-th frame exists), grouped into one launch --------------------
        if shape is None:               # dst owns no frame at all (n < dst + 1): learn the frame geometry from a probe
            probe = to_frame(render_fn(frames[0]), bits)
            shape, dtype, device = probe.shape, probe.dtype, probe.device
        ops, bufs = [], []
        for r in range(world):
            i = r + k * world
            if r == dst or i >= n:
                continue
            buf = torch.empty(shape, dtype=dtype, device=device)
            ops.append(dist.P2POp(dist.irecv, buf, r, group))
            bufs.append((i, buf))
        if q is not None:
            deliver([(mine[k], q)])
        if ops:
            pending.append((dist.batch_isend_irecv(ops), bufs))
        while len(pending) > 1:          # round k - 1 has had a whole render to arrive
            reqs, done = pending.pop(0)
            for r in reqs:
                r.wait()
            deliver(done)
    for reqs, done in pending:
        for r in reqs:
            r.wait()
        if rank == dst:
            deliver(done)
    if rank != dst:
        return None
    return None if on_frame is not None else out


def tile_row_plan(h_blocks, output_tile_step, out_tile_size, y_h, world):
    """Contiguous tile rows per rank for ``render_rows_sharded``: a list of ``(r0, r1, y0, y1)`` — rank k renders tile rows
    [r0, r1) and owns output rows [y0, y1).  Ranks beyond the number of tile rows get an empty share at the end."""
    overlap = out_tile_size - output_tile_step
    if overlap > output_tile_step:
        raise ValueError("tile-row sharding needs at most two tile rows over any output row")
    active = max(1, min(world, h_blocks))
    base, extra = divmod(h_blocks, active)
    plan, r = [], 0
    for k in range(world):
        n = (base + (1 if k < extra else 0)) if k < active else 0
        r0, r1 = r, r + n
        y0 = min(r0 * output_tile_step, y_h)
        y1 = y_h if (r1 >= h_blocks) else min(r1 * output_tile_step, y_h)
        if n == 0:
            y0 = y1 = y_h
        plan.append((r0, r1, y0, y1))
        r = r1
    return plan


def render_rows_sharded(x, eng, group=None, dst=0):
    """ONE image over the ranks of ``group`` by TILE ROWS (the fallback of SURVEY.md §8e for a single huge image; frames
    should be sharded whole with ``render_sharded``).  ``eng`` is a tile-row engine for this image on this rank's device
    (``model.row_engine(x)`` → ``SwinRowEngine``: ``render_tile_rows``, ``export_band`` / ``import_band``, ``stitch_rows`` and
    the grid numbers).  Every rank renders the tiles of its own tile rows; the ONLY exchange is the overlap band — the bottom
    ``out_tile_size - output_tile_step`` output rows of a rank's last tile row go to the next rank (one point-to-point
    message per boundary, RCCL over xGMI under ``nccl``), which imports them into its tile store and stitches its own band of
    output rows with the unchanged stitch kernel.  The bands are then delivered to ``dst``.  Tiles, recurrence order and
    arithmetic are those of the whole-frame render, so the result is bit-identical to it.  Returns CHW float on ``dst``,
    ``None`` elsewhere."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        eng.render_tile_rows(x, 0, eng.h_blocks)
        return eng.stitch_rows(0, eng.y_h)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    plan = tile_row_plan(eng.h_blocks, eng.output_tile_step, eng.out_tile_size, eng.y_h, world)
    r0, r1, y0, y1 = plan[rank]
    overlap = eng.out_tile_size - eng.output_tile_step
    reqs = []
    recv_band = None
    if r1 > r0 and r0 > 0 and overlap > 0:               # the previous rank's last tile row covers my first `overlap` rows
        recv_band = torch.empty((eng.w_blocks, 3, overlap, eng.out_tile_size), dtype=torch.float32, device=eng.device)
        reqs += dist.batch_isend_irecv([dist.P2POp(dist.irecv, recv_band, rank - 1, group)])
    if r1 > r0:
        eng.render_tile_rows(x, r0, r1)
        nxt = rank + 1
        if nxt < world and plan[nxt][1] > plan[nxt][0] and overlap > 0:
            send_band = eng.export_band(r1 - 1, eng.output_tile_step, overlap)
            reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, send_band, nxt, group)])
    for r in reqs:
        r.wait()
    if recv_band is not None:
        eng.import_band(r0 - 1, eng.output_tile_step, recv_band)
    band = eng.stitch_rows(y0, y1)
    # ---- deliver the bands to dst (they are disjoint row ranges of the output) ------------------------------------------
    if rank != dst:
        if y1 > y0:
            for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, band.contiguous(), dst, group)]):
                r.wait()
        return None
    out = torch.empty((band.shape[0], eng.y_h, band.shape[2]), dtype=band.dtype, device=band.device)
    out[:, y0:y1] = band
    ops, bufs = [], []
    for k, (_, _, a, b) in enumerate(plan):
        if k == dst or b <= a:
            continue
        buf = torch.empty((band.shape[0], b - a, band.shape[2]), dtype=band.dtype, device=band.device)
        ops.append(dist.P2POp(dist.irecv, buf, k, group))
        bufs.append((a, b, buf))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    for a, b, buf in bufs:
        out[:, a:b] = buf
    return out


class ConcurrentRenderer:
    """Frame-level concurrency INSIDE one GPU: ``n_streams`` replicas of a model (each with its own engine handle and
    workspace) on ``n_streams`` HIP streams; frames are dealt round-robin and come back in submission order.

    Why: every kernel of the tiled render is a chip-filling launch whose last workgroups leave most CUs idle (and the
    persistent kernels' static work split adds imbalance); with a second, independent frame in flight the scheduler
    fills those tails with the other frame's workgroups.  Measured on MI355X (bench.py, swin_unet 2x, 1080p): 254 ->
    271 MPix/s with 2 streams, 273 with 3.  This is the in-GPU half of what the reference's ``FrameCallbackPool``
    (nunif/utils/video.py:1622-1757) does with one Python thread per device replica; across GPUs frames are sharded by rank
    (``render_sharded``).  Ordering is by HIP events only — no host synchronisation until a result is read on the host.
    """

    def __init__(self, model_factory, n_streams=2, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ConcurrentRenderer needs a ROCm device; there is no CPU path")
        self.models = [model_factory().eval().to(self.device) for _ in range(max(1, n_streams))]
        self.streams = [torch.cuda.Stream(self.device) for _ in self.models]
        self._next = 0

    def submit(self, fn, *args, **kwargs):
        """Run ``fn(model, *args, **kwargs)`` on the next replica's stream; returns a handle for ``result``.  Inputs
        produced on the caller's current stream are ordered before the work."""
        k = self._next
        self._next = (k + 1) % len(self.models)
        st = self.streams[k]
        st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            out = fn(self.models[k], *args, **kwargs)
            ev = torch.cuda.Event()
            ev.record(st)
        return out, ev, st

    def result(self, handle):
        """Make the caller's current stream wait for the handle's work and return its tensor (still asynchronous)."""
        out, ev, _ = handle
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        if torch.is_tensor(out):
            out.record_stream(cur)
        return out

    def map(self, fn, items, depth=None):
        """Yield ``fn(model, item)`` for every item, in order, keeping ``depth`` (default: the stream count) in flight."""
        depth = depth or len(self.models)
        pending = []
        for it in items:
            pending.append(self.submit(fn, it))
            if len(pending) > depth:
                yield self.result(pending.pop(0))
        while pending:
            yield self.result(pending.pop(0))

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
