#!/usr/bin/env python3
"""Where does the periodic stall of the PCIe-inclusive path come from?  Times plain pinned-memory transfers of one 1080p frame
in (6.2 MB) and one 2x frame out (24.9 MB), no render: async memcpy vs zero-copy kernels, torch pinned vs hipHostRegister'ed
memory, per-iteration wall time over 80 iterations.  Env knobs (HSA_ENABLE_SDMA=0, ...) are applied by the caller."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd.iw3 import _ops  # noqa: E402

dev = torch.device("cuda:0")
H, W = 1080, 1920
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def registered(shape):
    n = int(np.prod(shape))
    arr = np.zeros(n + 4096, dtype=np.uint8)
    off = (-arr.ctypes.data) % 4096
    a = arr[off:off + n]
    rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), ctypes.c_uint(0))
    assert rc == 0, rc
    t = torch.from_numpy(a.reshape(shape))
    t._keep = arr
    return t


def stats(name, ts):
    ts = np.array(ts) * 1e3
    print(f"{name:46s} median {np.median(ts):7.2f} ms  p90 {np.percentile(ts, 90):7.2f}  max {ts.max():7.2f}  "
          f"n>3x median {(ts > 3 * np.median(ts)).sum():2d}/{len(ts)}  GB/s(median) {(H * W * 3 * 5) / np.median(ts) / 1e6:6.1f}")


def run(kind, mem):
    mk = (lambda s: torch.empty(s, dtype=torch.uint8).pin_memory()) if mem == "pinned" else registered
    h_in, h_out = mk((H, W, 3)), mk((2 * H, 2 * W, 3))
    if mem != "pinned":
        torch.Tensor.is_pinned = lambda self, *a, **k: True
    d_in = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    y = torch.rand(3, 2 * H, 2 * W, device=dev)
    d_out = torch.empty((2 * H, 2 * W, 3), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(dev)
    ts = []
    for i in range(80):
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            if kind == "memcpy":
                d_in.copy_(h_in, non_blocking=True)
                _ops.frame_to_tensor(d_in)
                _ops.to_frame(y, 8, out=d_out)
                h_out.copy_(d_out, non_blocking=True)
            elif kind == "zerocopy":
                _ops.frame_to_tensor(h_in, device=dev)
                _ops.to_frame(y, 8, out=h_out)
            elif kind == "h2d_only":
                d_in.copy_(h_in, non_blocking=True)
            elif kind == "d2h_only":
                h_out.copy_(d_out, non_blocking=True)
        st.synchronize()
        ts.append(time.perf_counter() - t0)
    stats(f"{kind} / {mem}", ts[5:])


def run_host_touch(mem, write_in, read_out, kind="zerocopy", with_render=False):
    """The same transfers, but the HOST also writes the input buffer before / reads the output buffer after each iteration
    (what a real frame loop does)."""
    mk = (lambda s: torch.empty(s, dtype=torch.uint8).pin_memory()) if mem == "pinned" else registered
    h_in, h_out = mk((H, W, 3)), mk((2 * H, 2 * W, 3))
    src = np.random.randint(0, 256, (H, W, 3), dtype=np.uint8)
    d_in = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    d_out = torch.empty((2 * H, 2 * W, 3), dtype=torch.uint8, device=dev)
    y = torch.rand(3, 2 * H, 2 * W, device=dev)
    big = torch.rand(64, 1024, 1024, device=dev)
    st = torch.cuda.Stream(dev)
    t_w, t_g, t_r = [], [], []
    for i in range(60):
        t0 = time.perf_counter()
        if write_in:
            np.copyto(h_in.numpy(), src)
        t1 = time.perf_counter()
        with torch.cuda.stream(st):
            if kind == "zerocopy":
                _ops.frame_to_tensor(h_in, device=dev)
            else:
                d_in.copy_(h_in, non_blocking=True)
                _ops.frame_to_tensor(d_in)
            if with_render:
                for _ in range(6):
                    big.mul_(1.0001)                   # ~7 ms of unrelated GPU work between the two edges
            if kind == "zerocopy":
                _ops.to_frame(y, 8, out=h_out)
            else:
                _ops.to_frame(y, 8, out=d_out)
                h_out.copy_(d_out, non_blocking=True)
        st.synchronize()
        t2 = time.perf_counter()
        if read_out:
            _ = int(h_out.numpy()[::64, ::64].sum())
        t3 = time.perf_counter()
        t_w.append(t1 - t0); t_g.append(t2 - t1); t_r.append(t3 - t2)
    tag = f"{kind}/{mem} write_in={int(write_in)} read_out={int(read_out)} render={int(with_render)}"
    stats(tag + " [host write]", t_w[5:])
    stats(tag + " [gpu+sync]", t_g[5:])
    stats(tag + " [host read]", t_r[5:])


if os.environ.get("PROBE", "basic") == "basic":
    for mem in ("pinned", "registered"):
        for kind in ("h2d_only", "d2h_only", "memcpy", "zerocopy"):
            try:
                run(kind, mem)
            except Exception as e:       # noqa: BLE001
                print(kind, mem, "failed:", e)
else:
    for kind in ("zerocopy", "memcpy"):
        run_host_touch("pinned", False, False, kind)
        run_host_touch("pinned", True, False, kind)
        run_host_touch("pinned", True, True, kind)
        run_host_touch("pinned", True, True, kind, with_render=True)
