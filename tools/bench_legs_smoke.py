#!/usr/bin/env python3
"""One-rank smoke test of bench.py's N > 1 legs on a real GPU: a single-process `nccl` group (world size 1) through
``iw3_sharded_leg``, ``cunet_sharded_leg`` and ``config5_replicas_leg`` with the HIP engine — the code paths the driver's 8-GPU run takes, minus the peers.
    python tools/bench_legs_smoke.py"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.set_grad_enabled(False)


def barrier():
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)


out = {"iw3": bench.iw3_sharded_leg(dist, 1, 0, dev, barrier, frames_per_rank=int(os.environ.get('LEG_FRAMES', '24'))),
       "cunet": bench.cunet_sharded_leg(dist, 1, 0, dev, barrier, frames_per_rank=4),
       "config5": bench.config5_replicas_leg(dist, 1, 0, dev, barrier),
       "multi_gpu": bench.collect_multi_gpu(dist, 1, 0, 0, dev)}
print(json.dumps(out))
dist.destroy_process_group()
