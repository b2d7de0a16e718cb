#!/usr/bin/env python3
"""Times the dominant kernel alone for several batch sizes (events on the
launch stream): shows how the block count quantises against the CU slots."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd.ops import PhotometricPlan  # noqa: E402
from depth_estimation_amd.synth import make_sequence  # noqa: E402

H, W = 192, 640
for B in [int(a) for a in sys.argv[1:]] or [8, 10, 12, 13, 16, 20, 24, 48]:
    seq = make_sequence(B, H, W, seed=0).to("cuda")
    plan = PhotometricPlan(B, H, W, 4, "cuda").bind(*seq.args())
    plan.forward(True)
    plan.tile_kernel_only()
    for _ in range(5):
        plan.forward(with_grad=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        plan.forward(with_grad=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B={B:3d} tile kernel {us:8.1f} us  {us / B:7.2f} us/sample", flush=True)
