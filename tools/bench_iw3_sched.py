#!/usr/bin/env python3
"""iw3 frame scheduler timing (BASELINE config 4 shape): uint8 1080p frames already in HBM -> FrameCallbackPool ->
bind_batch_frame_callback (Depth-Anything-V2 ViT-S stand-in, EMA min-max with a 4-frame look-ahead, a scene cut, stereo
method, SBS, quantise) with the depth stage and the stereo stage on two HIP streams vs on one.  Prints one JSON object.
Not the contract benchmark.

    python tools/bench_iw3_sched.py [--frames 48] [--batch 2] [--methods row_flow_v3,forward_fill]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd.iw3.base_depth_model import CallableDepthModel  # noqa: E402
from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2  # noqa: E402
from nunif_amd.iw3.frame_pipeline import FrameCallbackPool, PipelineOps, bind_batch_frame_callback  # noqa: E402
from nunif_amd.iw3.models.row_flow_v3 import RowFlowV3  # noqa: E402
from nunif_amd.iw3 import utils as U  # noqa: E402
from nunif_amd.synthetic import depth_anything_v2_state_dict, row_flow_v3_state_dict  # noqa: E402

DEV = "cuda:0"


class DeviceFrame:
    """A decoded frame that already sits in HBM as HWC uint8 (the bench keeps PCIe out of the timed region)."""

    def __init__(self, data, pts):
        self.data, self.pts = data, pts


def run(method, n_frames, batch, stage_streams, depth_model, side, frames):
    os.environ["NUNIF_IW3_STAGE_STREAMS"] = "1" if stage_streams else "0"
    args = argparse.Namespace(batch_size=batch, mapper="none", convergence=0.5, divergence=2.0, method=method,
                              synthetic_view="both", warp_steps=None, stereo_width=None, preserve_screen_border=False,
                              disable_amp=False, edge_dilation=2, pix_fmt="yuv420p", state={"device": torch.device(DEV)})
    depth_model.reset()
    depth_model.enable_ema(0.75, buffer_size=4)
    ops = PipelineOps(to_tensor=lambda frame, device=None: U.to_tensor(frame.data, device=device))
    cb, pre = bind_batch_frame_callback(depth_model, side, {n_frames // 2}, args, ops=ops)
    pool = FrameCallbackPool(frame_callback=cb, preprocess_callback=pre, batch_size=batch, device=DEV, max_workers=2,
                             max_batch_queue=3, require_pts=True, require_flush=True, ops=ops)
    n_out = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_frames):
        r = pool(DeviceFrame(frames[i % len(frames)], i))
        n_out += len(r) if r else 0
    n_out += len(pool(None))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert n_out == n_frames, (n_out, n_frames)
    return dt / n_frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--methods", default="row_flow_v3,forward_fill")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    H, W = 1080, 1920
    depth_model = CallableDepthModel(HipDepthAnythingV2(depth_anything_v2_state_dict(601), DEV))
    depth_model.load(gpu=0)
    side = RowFlowV3().eval()
    side.load_state_dict(row_flow_v3_state_dict(301))
    side = side.to(DEV)
    side.delta_output = True
    g = torch.Generator().manual_seed(5)
    frames = [torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).to(DEV) for _ in range(4)]
    res = {"frame": [H, W], "frames": a.frames, "batch": a.batch, "ema": [0.75, 4]}
    for method in a.methods.split(","):
        out = {}
        for streams in (False, True):
            run(method, 8, a.batch, streams, depth_model, side, frames)             # warm-up
            dt = min(run(method, a.frames, a.batch, streams, depth_model, side, frames) for _ in range(2))
            out["two_stage_streams" if streams else "one_stream"] = {
                "ms_per_frame": round(dt * 1e3, 3), "fps": round(1 / dt, 1), "input_MPix_s": round(H * W / dt / 1e6, 1)}
        res[method] = out
    print(json.dumps(res))


if __name__ == "__main__":
    main()
