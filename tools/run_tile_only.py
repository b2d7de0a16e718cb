#!/usr/bin/env python3
"""Launches only the dominant kernel (photo_tile_kernel<true>) a few times on
the benchmark workload; meant to be run under rocprofv3 --pmc."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd.ops import PhotometricPlan  # noqa: E402
from depth_estimation_amd.synth import make_sequence  # noqa: E402

B, H, W = int(os.environ.get("DW_B", 12)), int(os.environ.get("DW_H", 192)), int(os.environ.get("DW_W", 640))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
seq = make_sequence(B, H, W, seed=0).to("cuda")
plan = PhotometricPlan(B, H, W, 4, "cuda").bind(*seq.args())
plan.forward(with_grad=True)   # a complete pass first: the tile kernel reads the depth maps it leaves
plan.tile_kernel_only()
grad = os.environ.get("DW_GRAD", "1") != "0"   # DW_GRAD=0: the forward-only tile kernel
for _ in range(n):
    plan.forward(with_grad=grad)
torch.cuda.synchronize()
