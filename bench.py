#!/usr/bin/env python3
"""Benchmark of the nunif hot path on MI355X: waifu2x swin_unet 2x, tile 256, synthetic 1080p frames.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one full tiled render (gather -> 45 tiles through the net -> stitch) of one 1080p frame per rank, with
the input frame already resident in HBM.  Frames are independent, so ranks shard frames with no data-path
collective (weak scaling: every rank renders K frames).  ``value`` = input megapixels of all ranks / wall time
(max over ranks), the metric BASELINE.json names ("MPix" = source-frame pixels).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     — the dominant kernel (largest share of HIP-event time, classes are kernel symbols): algorithmic
                 FLOPs or bytes per launch / measured average launch duration, against the MI355X peak that bounds
                 it (arithmetic intensity vs the 2.5 PFLOP/s / 8 TB/s ridge); ``traffic`` = HBM bytes per launch
                 from the committed rocprofv3 PMC passes (profiles/, FETCH_SIZE x2 + WRITE_SIZE, KiB units)
  cpu_baseline — the CPU oracle (oracle/, a torch-fp32 port of the reference path) timed on the host cores on a
                 bounded sample of the same workload
"""
import argparse
import json
import math
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_H, FRAME_W = 1080, 1920
TILE = 256
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0       # spec (≈6.3 TB/s achievable)


def synth_frame(seed, h, w):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, h // 16 + 1, w // 16 + 1, generator=g)
    up = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0]
    return torch.clamp(up * 0.8 + 0.2 * torch.rand(3, h, w, generator=g), 0, 1)


def cpu_baseline(sd, frame):
    """Time the oracle's tiled_render on a crop of the same frame (bounded: ~10-30 s of CPU work)."""
    from oracle import seam_blending as OS
    from oracle import swin_unet as O
    crop = frame[:, :480, :480].contiguous()        # 3x3 tiles of 256 (step 236) — same tile size as the GPU run
    fn = lambda mb: O.model_forward(sd, mb)         # noqa: E731
    t0 = time.perf_counter()
    out = OS.tiled_render(crop, fn, 2, 16, 8, TILE, 4)
    dt = time.perf_counter() - t0
    return {"value": round(crop.shape[1] * crop.shape[2] / 1e6 / dt, 5), "unit": "MPix/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle tiled_render of a 480x480 crop (9 tiles of 256, batch 4), {dt:.1f} s"}, crop, out


def pmc_traffic_bytes(symbol):
    """HBM bytes per launch of ``symbol`` from the newest committed PMC summaries (profiles/rNN_pmc_*.txt)."""
    def per_launch(path):
        key = re.sub(r"[^A-Za-z0-9]", "", symbol.split("<")[0])
        args = re.findall(r"\d+", symbol.split("<", 1)[1]) if "<" in symbol else []
        for line in open(path):
            name = line.split(" launches=")[0]
            flat = re.sub(r"[^A-Za-z0-9]", "", name)
            # demangled "ns::kernel<6, 4>(…)" or mangled "_ZN5nunif15proj_mlp_kernelILi96ELi4EEE…"
            if key in flat and all((f"Li{a}E" in name) or re.search(rf"[<,]\s*{a}\s*[,>]", name) for a in args):
                m = re.search(r"per_launch=([0-9.]+)", line)
                if m:
                    return float(m.group(1)) * 1024.0
        return None
    pdir = os.path.join(ROOT, "profiles")
    rounds = sorted({re.match(r"(r\d\d[a-z]?)_pmc_FETCH_SIZE", f).group(1) for f in os.listdir(pdir)
                     if re.match(r"r\d\d[a-z]?_pmc_FETCH_SIZE", f)}) if os.path.isdir(pdir) else []
    for r in reversed(rounds):
        fetch = per_launch(os.path.join(pdir, f"{r}_pmc_FETCH_SIZE.txt"))
        write = per_launch(os.path.join(pdir, f"{r}_pmc_WRITE_SIZE.txt"))
        if fetch is not None and write is not None:
            return 2.0 * fetch + write, r      # FETCH_SIZE reports half of a wide coalesced read on gfx950
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=int(os.environ.get("NUNIF_BENCH_BATCH", "45")),
                    help="tiles per model launch (the reference's tile minibatch; results do not depend on it)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("NUNIF_BENCH_STREAMS", "2")),
                    help="frames rendered CONCURRENTLY per step, one HIP stream + one engine handle each (a step then "
                         "covers this many frames; the tails of one frame's kernels overlap the other frame's work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-frames", action="store_true",
                    help="skip the extra (reported, never `value`) PCIe-inclusive measurement through the pinned frame ring")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from nunif_amd import _hip
    from nunif_amd.nunif.utils.render import tiled_render
    from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x
    from nunif_amd.synthetic import swin_unet_state_dict    # seeded random-init weights (no checkpoints offline)

    torch.set_grad_enabled(False)
    sd = swin_unet_state_dict(102, 2)
    n_streams = max(1, args.streams)

    def make_model():
        mm = SwinUNet2x().eval()
        mm.load_state_dict(sd)
        return mm

    from nunif_amd.parallel import ConcurrentRenderer
    pool = ConcurrentRenderer(make_model, n_streams, dev)       # n_streams engine replicas, one HIP stream each
    model = pool.models[0]
    # a few distinct frames per rank, resident in HBM before the timed region
    frames = [synth_frame(1234 + rank * 16 + i, FRAME_H, FRAME_W).to(dev) for i in range(4)]

    def render(m, frame):
        return tiled_render(frame, m, tile_size=TILE, batch_size=args.batch_size)

    def step(i):
        if n_streams == 1:
            return render(model, frames[i % len(frames)])
        # one frame per stream, no cross-stream dependence; results are left on their streams (the barrier syncs them)
        return [pool.submit(render, frames[(i * n_streams + k) % len(frames)]) for k in range(n_streams)]

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same K frames on ONE stream (reported next to `value`; the per-kernel roofline below is measured this way) ----
    single = None
    if n_streams > 1 and rank == 0:
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(args.steps):
            render(model, frames[i % len(frames)])
        torch.cuda.synchronize(dev)
        dt1 = time.perf_counter() - t1
        single = {"value": round(FRAME_H * FRAME_W / 1e6 * args.steps / dt1, 2), "unit": "MPix/s",
                  "ms_per_frame": round(1e3 * dt1 / args.steps, 3), "frames": args.steps}

    # ---- per-kernel timing with HIP events on the launch stream (library hooks), outside the timed region ----------
    roofline, classes = None, []
    if rank == 0:
        _hip.profile_enable(True)
        n_prof = max(1, min(3, args.steps))
        for i in range(n_prof):             # per-kernel durations are measured on ONE stream (no overlap between frames)
            tiled_render(frames[i % len(frames)], model, tile_size=TILE, batch_size=args.batch_size)
        torch.cuda.synchronize(dev)
        recs = _hip.profile_read(reset=True)
        _hip.profile_enable(False)
        total = sum(r["total_ms"] for r in recs) or 1.0
        for r in sorted(recs, key=lambda r: -r["total_ms"]):
            sec = r["total_ms"] * 1e-3
            classes.append({"kernel": r["name"], "share": round(r["total_ms"] / total, 4),
                            "avg_us": round(1e3 * r["total_ms"] / max(1, r["launches"]), 2),
                            "launches_per_frame": r["launches"] // n_prof,
                            "tflops": round(r["flops"] / sec / 1e12, 2) if sec else 0.0,
                            "gbs": round(r["bytes"] / sec / 1e9, 1) if sec else 0.0})
        dom = max(recs, key=lambda r: r["total_ms"])
        n = max(1, dom["launches"])
        avg_s = dom["total_ms"] * 1e-3 / n
        flops_l, bytes_l = dom["flops"] / n, dom["bytes"] / n
        intensity = flops_l / bytes_l if bytes_l else float("inf")
        ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        traffic, src = pmc_traffic_bytes(dom["name"])
        if intensity >= ridge:
            ach = flops_l / avg_s / 1e12
            roofline = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4)}
        else:
            ach = bytes_l / avg_s / 1e9
            roofline = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        roofline.update({"traffic": traffic, "traffic_source": f"profiles/{src}_pmc_*" if src else None,
                         "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_flops_per_launch": flops_l,
                         "algorithmic_bytes_per_launch": bytes_l, "flop_per_byte": round(intensity, 1),
                         "achieved_tflops": round(flops_l / avg_s / 1e12, 2),
                         "mfma_frac": round(flops_l / avg_s / 1e12 / MFMA_PEAK_TFLOPS, 4)})

    if rank == 0:
        mpix_in = FRAME_H * FRAME_W / 1e6
        value = mpix_in * args.steps * n_streams * world / elapsed
        result = {
            "metric": "input MPix/s, waifu2x swin_unet 2x tiled render (tile 256) of 1080p frames",
            "value": round(value, 2), "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "waifu2x swin_unet 2x (art scale2x geometry), tile_size=256, 1080p frame, "
                                   "random-init weights", "frame": [FRAME_H, FRAME_W], "tile_size": TILE,
                       "tile_batch": args.batch_size, "tiles_per_frame": 45, "frames_per_step_per_gpu": n_streams,
                       "concurrent_streams": n_streams,
                       "parallelism": f"frame-sharded x{world}"},
            "output_mpix_per_s": round(value * 4, 2),
            "model_tflops": round(45 * 98e9 * args.steps * n_streams * world / elapsed / 1e12, 2),
            "model_mfma_frac": round(45 * 98e9 * args.steps * n_streams * world / elapsed / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
            "roofline": roofline, "kernel_classes": classes,
        }
        if single is not None:
            result["single_stream"] = single        # one frame at a time on one stream, same build, same run
            result["roofline"]["measured"] = "single-stream leg (kernel durations are not overlapped with another frame)"
        if not args.no_host_frames and world == 1:
            # host uint8 frame -> pinned ring -> H2D -> to_tensor -> render -> quantise -> D2H -> host uint8 frame
            from nunif_amd.frame_ring import FrameRing
            import numpy as np
            host = [(f.clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy() for f in frames]
            ring = FrameRing(lambda x: tiled_render(x, model, tile_size=TILE, batch_size=args.batch_size),
                             (FRAME_H, FRAME_W, 3), (2 * FRAME_H, 2 * FRAME_W, 3), device=dev, depth=3)
            for i in range(3):
                ring.submit(host[i % len(host)])
            ring.drain()
            n_host = max(8, args.steps)
            t1 = time.perf_counter()
            for i in range(n_host):
                ring.submit(host[i % len(host)])
            ring.drain()
            dt = time.perf_counter() - t1
            result["host_frames"] = {"mpix_per_s": round(mpix_in * n_host / dt, 2), "ms_per_frame": round(1e3 * dt / n_host, 3),
                                     "frames": n_host, "path": "uint8 HWC host -> pinned ring (depth 3) -> H2D -> render -> "
                                     "quantise -> D2H -> uint8 HWC host", "bytes_per_frame": int(FRAME_H * FRAME_W * 3 * 5)}
        if not args.no_cpu_baseline and world == 1:      # contract: CPU baseline on rank 0 at N = 1 only
            base, crop, ref = cpu_baseline(sd, frames[0].cpu())
            got = tiled_render(crop, model, tile_size=TILE, batch_size=args.batch_size).cpu()
            mse = torch.mean((got.double() - ref.double()) ** 2).item()
            result["cpu_baseline"] = base
            result["psnr_vs_oracle_db"] = round(10 * math.log10(1.0 / (mse + 1e-6)), 2)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
