#!/usr/bin/env python3
"""bench.py's config5 record on its own (4K batches of 3 -> depth wrapper -> mask-MLBW -> 12-frame queue -> video inpaint -> SBS):
the command the PMC passes of tools/profile_config5.sh run."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.set_grad_enabled(False)
rec = bench.config5_record(torch.device("cuda:0"))
print(json.dumps({k: rec[k] for k in ("ms_per_frame", "fps", "kernel_classes", "roofline") if k in rec}))
