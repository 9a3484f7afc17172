#!/usr/bin/env python3
"""iw3 per-frame pipeline timing (BASELINE config 4 shape): 1080p frame -> batch_preprocess -> Depth-Anything-V2 ViT-S
(random-init weights of the published architecture) -> min-max normalise -> stereo synthesis -> SBS compose + quantise.
Prints one JSON object; per-kernel classes come from the library's HIP-event profiler.  Not the contract benchmark."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunif_amd import _hip  # noqa: E402
from nunif_amd.iw3 import _ops  # noqa: E402
from nunif_amd.iw3.base_depth_model import CallableDepthModel  # noqa: E402
from nunif_amd.iw3.depth_anything_v2 import HipDepthAnythingV2  # noqa: E402
from nunif_amd.iw3.models.row_flow_v3 import RowFlowV3  # noqa: E402
from nunif_amd.iw3.utils import apply_divergence  # noqa: E402
from nunif_amd.iw3.mlbw_inpaint import MLBWInpaint  # noqa: E402
from nunif_amd.iw3.models.light_inpaint_v1 import LightInpaintV1  # noqa: E402
from nunif_amd.iw3.models.mlbw import MLBW  # noqa: E402
from nunif_amd.iw3.models.light_video_inpaint_v1 import LightVideoInpaintV1  # noqa: E402
from nunif_amd.iw3.video_depth_anything_streaming_model import VideoDepthAnythingStreamingModel  # noqa: E402
from nunif_amd.synthetic import (depth_anything_v2_state_dict, light_inpaint_state_dict, light_video_inpaint_state_dict,  # noqa: E402
                                 mlbw_state_dict, row_flow_v3_state_dict)

DEV = "cuda:0"


def main():
    torch.set_grad_enabled(False)
    H, W = 1080, 1920
    depth_model = CallableDepthModel(HipDepthAnythingV2(depth_anything_v2_state_dict(601), DEV))
    depth_model.load(gpu=0)
    side = RowFlowV3().eval()
    side.load_state_dict(row_flow_v3_state_dict(301))
    side = side.to(DEV)
    side.delta_output = True
    frames = [torch.rand(3, H, W, device=DEV) for _ in range(3)]
    res = {}
    for method in ("row_flow_v3", "forward_fill", "grid_sample"):
        args = SimpleNamespace(mapper="none", convergence=0.5, divergence=2.0, method=method, synthetic_view="both",
                               warp_steps=None, stereo_width=None, preserve_screen_border=False, disable_amp=False)

        def step(i):
            x = frames[i % 3]
            d = depth_model.infer(x, tta=False, edge_dilation=2)                 # [1,h,w] raw
            d = depth_model.minmax_normalize_chw(d)
            left, right = apply_divergence(d, x, args, side_model=side)
            return _ops.stereo_to_frame(left, right, "sbs")

        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[method] = {"ms_per_frame": round(dt * 1e3, 3), "fps": round(1 / dt, 1), "input_MPix_s": round(H * W / dt / 1e6, 1)}
    # image-mode MLBW + inpaint (config 5 without the temporal queue): mask MLBW warp, hole masks, light_inpaint_v1 per eye
    inp = LightInpaintV1().eval()
    inp.load_state_dict(light_inpaint_state_dict(701))
    mm = MLBW(num_layers=2, base_dim=32, hole_mask=True).eval()
    mm.load_state_dict(mlbw_state_dict(431, 2, False, hole_mask=True))
    inpaint = MLBWInpaint(inp.to(DEV), mm.to(DEV))

    def step_inpaint(i):
        x = frames[i % 3]
        d = depth_model.minmax_normalize_chw(depth_model.infer(x, tta=False, edge_dilation=2))
        left, right = inpaint.infer(x[None], d[None], divergence=2.0, convergence=0.5, synthetic_view="both",
                                    inner_dilation=1, outer_dilation=1)
        return _ops.stereo_to_frame(left[0], right[0], "sbs")

    for i in range(2):
        step_inpaint(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        step_inpaint(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    res["mlbw_inpaint_image"] = {"ms_per_frame": round(dt * 1e3, 3), "fps": round(1 / dt, 1),
                                 "input_MPix_s": round(H * W / dt / 1e6, 1)}
    # ---- BASELINE config 5 shape: 4K frames, VideoDepthAnythingStreaming wrapper (per-frame ViT-S stand-in for the external
    #      streaming net) -> min-max -> mask-MLBW warp -> 12-frame queue -> light_video_inpaint_v1 on both eyes -> SBS u8 ----
    if "--no-cfg5" not in sys.argv:
        H5, W5 = 2160, 3840
        vda = VideoDepthAnythingStreamingModel("VDA_Stream_S", backbone=depth_model.model).load(gpu=0)
        vid = LightVideoInpaintV1().eval()
        vid.load_state_dict(light_video_inpaint_state_dict(801))
        inpaint_v = MLBWInpaint(inp.to(DEV), mm.to(DEV), video_model=vid.to(DEV))
        inpaint_v.set_mode("video")
        frames5 = [torch.rand(3, H5, W5, device=DEV) for _ in range(3)]
        n_out = [0]

        def step5(i, batch=3):
            x = torch.stack([frames5[(i + k) % 3] for k in range(batch)])
            d = vda.infer(x, edge_dilation=2)                                     # [B,1,h,w] raw
            d = torch.stack(vda.minmax_normalize(d))
            left, right = inpaint_v.infer(x, d, divergence=2.0, convergence=0.5, synthetic_view="both",
                                          inner_dilation=1, outer_dilation=1)
            if left is None:                      # the 12-frame queue is still filling
                return
            for k in range(left.shape[0]):
                _ops.stereo_to_frame(left[k], right[k], "sbs")
                n_out[0] += 1

        for i in range(4):                       # fills the 12-frame queue (first outputs appear after 3 calls)
            step5(i)
        torch.cuda.synchronize()
        n_out[0] = 0
        t0 = time.perf_counter()
        n_in = 0
        for i in range(8):
            step5(i)
            n_in += 3
        torch.cuda.synchronize()
        dt5 = time.perf_counter() - t0
        res["cfg5_4k_vda_stream_mlbw_video_inpaint"] = {
            "frames_in": n_in, "frames_out": n_out[0], "ms_per_frame": round(1e3 * dt5 / n_in, 2), "fps": round(n_in / dt5, 1),
            "input_MPix_s": round(H5 * W5 * n_in / dt5 / 1e6, 1),
            "note": "streaming depth net = per-frame ViT-S stand-in (the external temporal head is not restated)"}
        inpaint_v.reset()
    _hip.profile_enable(True)
    step_inpaint(0)
    torch.cuda.synchronize()
    recs = _hip.profile_read(reset=True)
    res["kernel_classes_inpaint"] = [{"kernel": r["name"], "us": round(r["total_ms"] * 1e3, 1), "launches": r["launches"]}
                                     for r in sorted(recs, key=lambda r: -r["total_ms"])[:16]]
    step(0)
    torch.cuda.synchronize()
    recs = _hip.profile_read()
    _hip.profile_enable(False)
    tot = sum(r["total_ms"] for r in recs)
    res["kernel_classes_last_method"] = [{"kernel": r["name"], "us": round(r["total_ms"] * 1e3, 1), "launches": r["launches"]}
                                         for r in sorted(recs, key=lambda r: -r["total_ms"])[:14]]
    res["gpu_ms_profiled"] = round(tot, 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
