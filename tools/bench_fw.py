#!/usr/bin/env python3
"""The forward-warp workload of bench.py's iw3 record on its own (forward_fill, both eyes, 2 x 1080p per launch, divergence
2.0) + dilate_edge(2) on a 2 x 392 x 686 depth map: the command the rocprofv3 PMC passes of tools/profile_iw3_ops.sh run, so
that FETCH_SIZE / WRITE_SIZE per launch belong to exactly the launch shape bench.py prices."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_frame  # noqa: E402
from nunif_amd.iw3.dilation import dilate_edge  # noqa: E402
from nunif_amd.iw3.forward_warp import apply_divergence_forward_warp  # noqa: E402
from nunif_amd.synthetic import synth_depth  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
c = torch.stack([synth_frame(910 + i, 1080, 1920) for i in range(2)]).to(dev)
d = synth_depth(1, 2, 1080, 1920, "smooth_edges").to(dev)
ds = synth_depth(2, 2, 392, 686, "smooth_edges").to(dev) * 5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    apply_divergence_forward_warp(c, d, 2.0, 0.5, method="forward_fill", width_base=False)
    dilate_edge(ds, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    apply_divergence_forward_warp(c, d, 2.0, 0.5, method="forward_fill", width_base=False)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    dilate_edge(ds, 2)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"forward_fill 2x1080p: {(t1 - t0) / n * 1e6:.1f} us per launch = {2 * 1080 * 1920 * 40 / ((t1 - t0) / n) / 1e12:.2f} TB/s algorithmic;"
      f" dilate_edge(2): {(t2 - t1) / n * 1e6:.1f} us per call")
