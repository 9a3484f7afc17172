"""A stand-in for the reference's ``iw3/cli.py`` on the VIDEO batch route, for ``tests/test_launch.py``
(``NUNIF_AMD_LAUNCH_CLI_MODULE=fake_cli.iw3_cli``).  It resolves — at call time, through the LIVE reference modules, with the
reference's own keyword arguments (``iw3/utils.py:1137-1153``) — the two names the launcher rebinds for one video on N GPUs,
``iw3.utils.bind_batch_frame_callback`` and ``VU.FrameCallbackPool``, and then runs the decode loop of
``VU.process_video`` (``nunif/utils/video.py:1081-1127``: one callback call per decoded frame, one with ``None`` at the end) over
synthetic frames.  Depth net, warp and "encoder" are CPU stand-ins (the frame-pool fixture's: ``tests/golden/make_golden_cases.py``),
so rank 0's output must be the fixture's frames, bit for bit."""
import argparse
import json
import os
import sys
import types

import torch

from oracle import refstub

refstub.install()

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden"))
from make_golden_cases import FRAME_POOL_CASES, fake_depth_net, frame_pool_frames  # noqa: E402

import iw3.utils as IU  # noqa: E402
import nunif.utils.video as VU  # noqa: E402

from nunif_amd.iw3 import frame_pipeline as FP  # noqa: E402
from nunif_amd.iw3.base_depth_model import BaseDepthModel  # noqa: E402
from oracle.backward_warp import grid_sample_warp  # noqa: E402


class _Frame:
    def __init__(self, x, pts):
        self.x, self.pts = x, pts


class _FakeDepth(BaseDepthModel):
    def load_model(self, model_type, resolution=None, device=None, **kw):
        return None

    def is_metric(self):
        return False

    def infer(self, x, **kw):
        return fake_depth_net(x) if x.ndim == 4 else fake_depth_net(x[None])[0]


class _AvFrame:
    """What ``av.VideoFrame.from_ndarray`` hands the encoder, as far as this test cares: the pixels."""

    def __init__(self, arr, fmt):
        self.arr, self.format = arr, fmt

    @classmethod
    def from_ndarray(cls, arr, format="rgb24"):
        return cls(arr, format)


def _cpu_ops(real_ops_cls):
    def apply_divergence(depths, x, args, side_model=None, reset_pts=None):
        return grid_sample_warp(x, depths, args.divergence, args.convergence, args.synthetic_view)

    return real_ops_cls(to_tensor=lambda frame, device=None: frame.x, preprocess_image=lambda x, args: x,
                        apply_divergence=apply_divergence,
                        postprocess_image=lambda le, re, args: torch.clamp(torch.cat([le, re], dim=2), 0, 1),
                        to_frame=lambda x, use_16bit=False: x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", "-i", required=True)
    ap.add_argument("--output", "-o", required=True)
    ap.add_argument("--gpu", "-g", type=int, nargs="+", default=[0])
    ap.add_argument("--case", default="ema")
    # the route through the reference's own ``process_video`` (iw3/utils.py:1209-1225) and the options its early exits read
    ap.add_argument("--via-process-video", action="store_true")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--yes", "-y", action="store_true")
    ap.add_argument("--low-vram", action="store_true")
    a = ap.parse_args()
    n, bs, cuts, ema, _ = FRAME_POOL_CASES[a.case]
    rank, world = int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # CPU stand-ins for the device functions (the HIP ops refuse CPU tensors) and for PyAV's frame type
    real_ops = FP.PipelineOps
    FP.PipelineOps = lambda *aa, **kw: _cpu_ops(real_ops) if not aa and not kw else real_ops(*aa, **kw)
    sys.modules["av"] = types.SimpleNamespace(VideoFrame=_AvFrame)
    args = argparse.Namespace(batch_size=bs, tta=False, low_vram=a.low_vram, disable_amp=True, edge_dilation=0, depth_aa=False,
                              rgbd=False, half_rgbd=False, method="grid_sample", mapper="none", divergence=2.0, convergence=0.5,
                              synthetic_view="both", pix_fmt="yuv420p", max_workers=0,
                              # what make_output_filename / process_video / the early exits read (iw3/utils.py:111-168,974-1011)
                              vr180=False, half_sbs=False, tb=False, half_tb=False, cross_eyed=False, anaglyph=None,
                              debug_depth=False, metadata=None, video_extension=".mp4", keyframe=False, resume=a.resume,
                              skip_error=False, yes=a.yes,
                              state={"device": torch.device("cpu"), "devices": [torch.device("cpu")], "convergence_model": None})
    depth_model = _FakeDepth("fake")

    def decode_and_encode(input_filename, output_path, args, depth_model, side_model):
        if ema is not None:
            depth_model.enable_ema(ema[0], buffer_size=ema[1])
        # ---- iw3/utils.py:1134-1153, verbatim in what it resolves and passes --------------------------------------------------
        extra_queue = 1 if len(args.state["devices"]) == 1 else 0
        minibatch_size = args.batch_size // 2 or 1 if args.tta else args.batch_size
        frame_callback, preprocess_callback = IU.bind_batch_frame_callback(
            depth_model=depth_model, side_model=None, segment_pts=set(cuts), args=args)
        pool_cls = VU.FrameCallbackPool
        frame_callback = pool_cls(frame_callback=frame_callback, preprocess_callback=preprocess_callback, batch_size=minibatch_size,
                                  device=args.state["devices"], max_workers=args.max_workers,
                                  max_batch_queue=args.max_workers + extra_queue, require_pts=True, require_flush=True, use_16bit=False)
        # ---- nunif/utils/video.py:1081-1127: the decode loop ----------------------------------------------------------------
        encoded, calls = [], []
        for i, x in enumerate(frame_pool_frames(n)):
            got = frame_callback(_Frame(x, i)) or []
            calls.append(len(got))
            encoded += got
        got = frame_callback(None) or []
        calls.append(len(got))
        encoded += got
        frame_callback.shutdown()
        os.makedirs(output_path, exist_ok=True)
        if encoded:
            torch.save(torch.stack([torch.from_numpy(f.arr) for f in encoded]), os.path.join(output_path, "frames.pt"))
        with open(os.path.join(output_path, f"rank{rank}.json"), "w") as f:
            json.dump({"rank": rank, "world": world, "gpu": a.gpu, "frames_encoded": len(encoded), "calls": calls,
                       "pool": type(frame_callback).__module__ + "." + type(frame_callback).__name__,
                       "av_frames": all(isinstance(e, _AvFrame) for e in encoded)}, f)

    if a.via_process_video:
        # the reference's process_video (resolved at call time: the launcher's guard wraps it) around a stand-in for the body of
        # process_video_full; what it was asked to do is recorded beside the output so that the test can see who ran what
        ran = []

        def body(input_filename, output_path, args, depth_model, side_model):
            ran.append(output_path)
            if args.low_vram:                 # the per-frame route: no pool, no collectives — one rank must be alone in here
                return None
            # the "video" is an (empty) FILE at output_path, as the early exits test it; the frames go beside it
            decode_and_encode(input_filename, output_path + ".frames", args, depth_model, side_model)
            if rank == 0:
                open(output_path, "wb").close()
            return None

        IU.process_video_full = body
        IU.process_video(a.input, a.output, args, depth_model, None)
        trace = os.environ.get("NUNIF_AMD_FAKE_CLI_TRACE")
        if trace:
            with open(os.path.join(trace, f"trace{rank}.json"), "w") as f:
                json.dump({"rank": rank, "ran": ran}, f)
    else:
        decode_and_encode(a.input, a.output, args, depth_model, None)
