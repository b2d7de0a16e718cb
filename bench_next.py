#!/usr/bin/env python3
"""Secondary benchmarks for the stages next to the headline path (bench.py is
the headline).  One JSON line per kernel, same fields as bench.py.

    python bench_next.py [--steps K] [--warmup W]

soft_argmin: the disparity regression at the end of DepthBlock.forward
(/root/reference/model/model.py:417-429, 454-456) on a (B, 256, 192, 640)
float32 cost volume, B = 4 (503 MB: larger than the 256 MiB Infinity Cache, so
the reads come from HBM).  Algorithmic bytes: forward D*4 + 4 per pixel,
backward 2*D*4 + 4 per pixel.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    from depth_estimation_amd._lib import lib
    torch.cuda.set_device(0)
    B, D, H, W = args.batch, 256, args.height, args.width
    N = H * W
    cost = torch.randn(B, D, H, W, device="cuda")
    disp = torch.empty(B, 1, H, W, device="cuda")
    g_disp = torch.randn(B, 1, H, W, device="cuda")
    g_cost = torch.empty_like(cost)
    st = torch.cuda.current_stream().cuda_stream
    L = lib()

    def fwd():
        assert L.dw_soft_argmin_fwd(B, D, N, cost.data_ptr(), disp.data_ptr(), st) == 0

    def bwd():
        assert L.dw_soft_argmin_bwd(B, D, N, cost.data_ptr(), g_disp.data_ptr(), g_cost.data_ptr(), st) == 0

    for name, fn, nbytes in (("soft_argmin_fwd", fwd, B * N * (D * 4 + 4)),
                             ("soft_argmin_bwd", bwd, B * N * (2 * D * 4 + 4))):
        ms = timed(fn, args.steps, args.warmup)
        gbps = nbytes / (ms * 1e-3) / 1e9
        out = {"metric": f"{name} throughput", "value": B / (ms * 1e-3), "unit": "volumes/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"soft-argmin over a ({B},{D},{H},{W}) cost volume"},
               "roofline": {"bound": "hbm", "kernel": "soft_argmin_kernel", "achieved": gbps,
                            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                            "traffic": None, "algorithmic_bytes": nbytes}}
        if name == "soft_argmin_fwd" and not args.no_cpu_baseline:
            from oracle.soft_argmin import soft_argmin as ref
            c = cost[:1].cpu()
            ref(c)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 10 and n < 10:
                ref(c)
                n += 1
            per = (time.perf_counter() - t0) / n
            out["cpu_baseline"] = {"value": 1 / per, "unit": "volumes/s", "cores": torch.get_num_threads(),
                                   "kind": "port", "sample": f"{n} forward passes of one (1,{D},{H},{W}) volume"}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
