"""Host <-> device frame ring: pinned buffers + copy streams so that H2D, compute and D2H of consecutive frames overlap.

Replaces what the reference does per frame with blocking calls: ``VU.to_tensor`` (``nunif/utils/video.py:218-223``) =
``torch.from_numpy(frame).to(device)``, the model call, then ``VU.to_frame`` / ``.cpu().numpy()`` (:236-269) inside a
``FrameCallbackPool`` worker thread (:1622-1757).  Here one thread drives ``depth`` slots:

    slot k:   [pinned u8 in] --copy stream--> [device u8] --compute stream: frame_to_tensor, process_fn,
              stereo_to_frame / quantise--> [device u8 out] --copy-back stream--> [pinned u8 out]

Ordering is by HIP events only (no host sync until a result is consumed), results come back in submission order.
PyTorch owns the memory and the stream; the conversions at both ends are the HIP kernels of iw3_frame.hip, which by
default read / write the pinned host buffers directly (zero-copy over PCIe) instead of going through async memcpys.
"""
import collections

import numpy as np
import torch

from .iw3 import _ops


class FrameRing:
    def __init__(self, process_fn, in_shape, out_shape, device="cuda:0", depth=3, bits=8, zero_copy=True, out_mode="copy",
                 edge_streams=True):
        """``process_fn(chw_float_tensor) -> chw_float_tensor in [0,1]`` runs on the ring's stream.
        in_shape / out_shape: (H, W, 3) of the uint8 (or uint16 when bits=16) HWC frames.

        zero_copy=True (default): the two edge kernels read / write the PINNED host buffers directly over PCIe
        (``frame_to_tensor`` gathers from host memory, ``to_frame`` scatters the quantised frame into it) — no
        ``hipMemcpyAsync`` at all.  Measured on MI355X / ROCm 7.2 (tools/ring_probe.py): async copies, on the compute
        stream or on their own, blocked the submitting thread for a whole frame time (5-7 ms inside ``copy_``) and
        the pipeline ran at 19-20 ms per 1080p 2x frame against 9.5 ms of GPU work."""
        assert bits in (8, 16) and out_mode in ("copy", "view")
        # out_mode: what a finished frame is handed back as.  "copy" = a fresh numpy array (safe to keep; costs one
        # single-threaded 25 MB host memcpy + a page-faulting allocation per 1080p 2x frame, which — not PCIe — was the
        # 21 ms / frame of round 1: tools/pcie_probe.py moves the same bytes over PCIe in 0.6 ms with no stalls at all);
        # "view" = a pinned output buffer itself, valid UNTIL THE NEXT CALL THAT RETURNS A FRAME (submit / input_buffer /
        # drain): the ring owns depth + 1 output buffers and a collected buffer is parked as the spare — nothing is queued
        # into it — until the next collect hands it to the slot that is being re-armed.  (Round 2 returned the slot's own
        # buffer and re-armed that slot in the same call: the GPU overwrote the view depth frames later with no host call
        # in between — the advisor's torn-frame race.)
        self.out_mode = out_mode
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FrameRing needs a ROCm device; there is no CPU path")
        self.process_fn, self.bits, self.depth, self.zero_copy = process_fn, bits, depth, zero_copy
        t_dtype = torch.uint8 if bits == 8 else torch.int16          # int16 carries the uint16 bit pattern
        self.slots = []
        for _ in range(depth):
            self.slots.append({
                "h_in": torch.empty(in_shape, dtype=t_dtype).pin_memory(),
                "h_out": torch.empty(out_shape, dtype=t_dtype).pin_memory(),
                "d_in": None if zero_copy else torch.empty(in_shape, dtype=t_dtype, device=self.device),
                "done": torch.cuda.Event(),
                "busy": False,
            })
        self.stream = torch.cuda.Stream(self.device)
        self.edge_streams = edge_streams
        self.in_stream = torch.cuda.Stream(self.device) if edge_streams else None
        self.out_stream = torch.cuda.Stream(self.device) if edge_streams else None
        self._spare = torch.empty(out_shape, dtype=t_dtype).pin_memory()     # see out_mode "view"
        self._next = 0
        self._pending = collections.deque()

    def submit(self, frame):
        """Queue one HWC uint8/uint16 numpy frame.  Returns a finished output frame (numpy, a copy) when the ring was
        full, else None.  Call ``drain()`` at the end."""
        out = getattr(self, "_held", None)
        self._held = None
        slot = self.slots[self._next]
        if slot["busy"]:
            out = self._collect()
        slot = self.slots[self._next]
        self._next = (self._next + 1) % self.depth
        if frame is not None:                                          # None: the caller filled input_buffer() itself
            # host memcpy into the pinned buffer — with numpy, NOT ``Tensor.copy_``: torch's copy into a pinned tensor took
            # 0.14 ms at the median but 70-90 ms every 3rd-4th frame (tools/ring_trace.py: 13 of round 1's 21 ms per frame;
            # the "platform stall" of DESIGN.md round 1 was this call), np.copyto is a flat 0.13 ms
            np.copyto(slot["h_in"].numpy(), frame.view(np.int16) if self.bits == 16 else frame)
        if self.zero_copy and self.edge_streams:
            # three streams: the PCIe-bound edge kernels of neighbouring frames (0.12 ms in, 0.45 ms out for 1080p -> 2x) run
            # beside the render of the frame in the middle instead of in front of / behind it on one stream
            with torch.cuda.stream(self.in_stream):
                x = _ops.frame_to_tensor(slot["h_in"], device=self.device)
                ev_in = torch.cuda.Event()
                ev_in.record(self.in_stream)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev_in)
                x.record_stream(self.stream)
                y = self.process_fn(x)
                ev_c = torch.cuda.Event()
                ev_c.record(self.stream)
            with torch.cuda.stream(self.out_stream):
                self.out_stream.wait_event(ev_c)
                y.record_stream(self.out_stream)
                _ops.to_frame(y, self.bits, out=slot["h_out"])
                slot["done"].record(self.out_stream)
            slot["busy"] = True
            self._pending.append(slot)
            return out
        with torch.cuda.stream(self.stream):
            if self.zero_copy:
                x = _ops.frame_to_tensor(slot["h_in"], device=self.device)
                y = self.process_fn(x)
                _ops.to_frame(y, self.bits, out=slot["h_out"])
            else:
                slot["d_in"].copy_(slot["h_in"], non_blocking=True)
                y = self.process_fn(_ops.frame_to_tensor(slot["d_in"]))
                slot["h_out"].copy_(_ops.to_frame(y, self.bits), non_blocking=True)
            slot["done"].record(self.stream)
        slot["busy"] = True
        self._pending.append(slot)
        return out

    def _collect(self):
        slot = self._pending.popleft()
        slot["done"].synchronize()
        buf = slot["h_out"]
        slot["h_out"], self._spare = self._spare, buf      # the consumer's buffer leaves the rotation until the next collect
        arr = buf.numpy()
        out = arr.view(np.uint16) if self.bits == 16 else arr
        if self.out_mode == "copy":
            out = out.copy()
        slot["busy"] = False
        return out

    def input_buffer(self):
        """The pinned input buffer of the NEXT slot as a numpy view (HWC): a decoder can write the frame straight into it
        and call ``submit(None)`` — saves the host memcpy of ``submit(frame)``.  Collects the oldest frame first when the
        ring is full; that frame is returned by the following ``submit``."""
        slot = self.slots[self._next]
        if slot["busy"]:
            self._held = self._collect()
        arr = self.slots[self._next]["h_in"].numpy()
        return arr.view(np.uint16) if self.bits == 16 else arr

    def drain(self):
        outs = []
        while self._pending:
            outs.append(self._collect())
        return outs
