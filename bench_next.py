#!/usr/bin/env python3
"""Secondary benchmarks for the stages next to the headline path (bench.py is
the headline).  One JSON line per kernel, same fields as bench.py.

    python bench_next.py [--steps K] [--warmup W]

soft_argmin: the disparity regression at the end of DepthBlock.forward
(/root/reference/model/model.py:417-429, 454-456) on a (B, 256, 192, 640)
float32 cost volume, B = 4 (503 MB: larger than the 256 MiB Infinity Cache, so
the reads come from HBM).  Algorithmic bytes: forward D*4 + 4 per pixel,
backward 2*D*4 + 4 per pixel.

cost_volume: DepthBlock.cost_volume (model.py:460-508) at the feature-map size
of a 192 x 640 input (F = 32, h = 96, w = 320, D = 64), B = 2: two
(B, 64, 128, 96, 320) volumes = 2 x 1.0 GB per sample.  Algorithmic bytes:
forward = the two volumes written + the four maps read; backward = the two
volume gradients read + the four map gradients written.
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
    ap.add_argument("--only", choices=["compute_loss", "soft_argmin", "cost_volume", "metrics", "lidar"], default=None)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    if args.only in (None, "compute_loss"):
        bench_compute_loss(args)
    if args.only in (None, "soft_argmin"):
        bench_soft_argmin(args)
    if args.only in (None, "cost_volume"):
        bench_cost_volume(args)
    if args.only in (None, "metrics"):
        bench_metrics(args)
    if args.only in (None, "lidar"):
        bench_lidar(args)


def line(name, unit, units, ms, args, workload, kernel, nbytes):
    gbps = nbytes / (ms * 1e-3) / 1e9
    return {"metric": f"{name} throughput", "value": units / (ms * 1e-3), "unit": unit, "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": gbps,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                         "traffic": None, "algorithmic_bytes": nbytes}}


def cpu_time(fn, budget_s=10.0, max_n=10):
    fn()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s and n < max_n:
        fn()
        n += 1
    return (time.perf_counter() - t0) / n, n


def bench_compute_loss(args):
    """Train.compute_loss (/root/reference/train.py:272-309): a 375 x 1242 ground-truth map resized to
    256 x 512 (the reference's own sizes, config.py:20-21) against the prediction.  One sample is what
    the reference evaluates per call (twice per step, left and right: train.py:113); the batched lines
    are the same call over B samples.  Algorithmic bytes: both maps read once."""
    from depth_estimation_amd._lib import lib
    L = lib()
    for B, Ht, Wt, H, W in ((1, 375, 1242, 256, 512), (64, 375, 1242, 256, 512), (256, 375, 1242, 256, 512),
                            (64, args.height, args.width, args.height, args.width)):
        target = torch.rand(B, Ht, Wt, device="cuda")
        pred = torch.rand(B, H, W, device="cuda")
        out = torch.empty(B, 4, device="cuda")
        nb = L.dw_compute_loss_workspace_bytes(B, H, W)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")

        def run():  # on the current stream: the graph capture below runs on a stream of its own
            assert L.dw_compute_loss_fwd(B, Ht, Wt, H, W, target.data_ptr(), pred.data_ptr(), out.data_ptr(), 0,
                                         ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream) == 0

        run()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
        ms = min(timed(run, args.steps, args.warmup), timed(graph.replay, args.steps, args.warmup))
        nbytes = B * 4 * (Ht * Wt + H * W)
        o = line("compute_loss", "samples/s", B, ms, args,
                 f"compute_loss of ({B},{Ht},{Wt}) ground truth against ({B},{H},{W}) predictions",
                 "compute_loss_strip_kernel", nbytes)
        if B == 1:
            # the call as the reference makes it (train.py:113): float64 ground truth, Python entry
            # point, result left on the device; wall clock per call, host side included
            from depth_estimation_amd.loss import compute_loss
            t64 = target.double()
            for _ in range(20):
                compute_loss(t64, pred)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                compute_loss(t64, pred)
            torch.cuda.synchronize()
            o["python_call_us"] = 1e6 * (time.perf_counter() - t0) / 200
        if B == 1 and not args.no_cpu_baseline:
            from oracle.compute_loss import compute_loss as ref
            t1, p1 = target.cpu(), pred.cpu()
            per, n = cpu_time(lambda: ref(t1, p1), max_n=50)
            o["cpu_baseline"] = {"value": 1 / per, "unit": "samples/s", "cores": torch.get_num_threads(),
                                 "kind": "port", "sample": f"{n} evaluations of one sample"}
        print(json.dumps(o), flush=True)


def bench_metrics(args):
    """Loss + rmse + extrema in one pass (train.py:272-343, 250-251), then normalize_image."""
    from depth_estimation_amd._lib import lib
    B, H, W = 12, args.height, args.width
    gt = torch.rand(B, 1, H, W, device="cuda") * 80
    pred = torch.rand(B, 1, H, W, device="cuda") + 1e-3
    out = torch.empty(B, 8, device="cuda")
    img = torch.empty_like(pred)
    L = lib()
    nb = L.dw_compute_loss_workspace_bytes(B, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def fused():   # two launches: the strip kernel, then the sums and the normalised image
        assert L.dw_compute_loss_metrics_norm_fwd(B, H, W, H, W, gt.data_ptr(), pred.data_ptr(), 1e-3, 128.0,
                                                  out.data_ptr(), img.data_ptr(), 0, ws.data_ptr(), nb,
                                                  torch.cuda.current_stream().cuda_stream) == 0

    fused()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fused()
    ms = min(timed(fused, args.steps, args.warmup), timed(graph.replay, args.steps, args.warmup))
    nbytes = B * H * W * (8 + 8)   # loss+metrics: two maps in; normalize: one in, one out
    o = line("loss_and_metrics", "samples/s", B, ms, args,
             f"compute_loss + rmse + extrema + normalize_image on ({B},1,{H},{W})", "compute_loss_strip_kernel",
             nbytes)
    if not args.no_cpu_baseline:
        from oracle.compute_loss import compute_loss_terms
        from oracle.metrics import compute_errors_one_side, normalize_image
        g1, p1 = gt[:1].cpu(), pred[:1].cpu()

        def ref():
            compute_loss_terms(g1, p1)
            compute_errors_one_side(g1, p1[:, 0], H, W)
            normalize_image(p1)
        per, n = cpu_time(ref, max_n=20)
        o["cpu_baseline"] = {"value": 1 / per, "unit": "samples/s", "cores": torch.get_num_threads(),
                             "kind": "port", "sample": f"{n} evaluations of one sample"}
    print(json.dumps(o), flush=True)


def bench_lidar(args):
    """Velodyne projection + sparse depth map (load_dataset.py:116-148), 64 scans of 120k points."""
    import numpy as np
    from depth_estimation_amd._lib import lib
    from depth_estimation_amd.lidar import KITTI_LIKE_CALIB, synthetic_scan, velo_to_image_matrix
    B, n, H, W = 64, 120000, 375, 1242
    P = velo_to_image_matrix(KITTI_LIKE_CALIB["cam2cam"], KITTI_LIKE_CALIB["velo2cam"], 2)
    scan = synthetic_scan(n, seed=0)
    pts = torch.from_numpy(scan)[None].repeat(B, 1, 1).cuda()
    npts = torch.full((B,), n, dtype=torch.int32, device="cuda")
    Pb = torch.from_numpy(P)[None].repeat(B, 1, 1).cuda()
    pts_im = torch.zeros(B, n, 3, dtype=torch.float64, device="cuda")
    n_valid = torch.empty(B, dtype=torch.int32, device="cuda")
    depth = torch.empty(B, H, W, dtype=torch.float64, device="cuda")
    L = lib()
    nb = L.dw_lidar_workspace_bytes(B, n, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run():
        assert L.dw_lidar_project(B, n, H, W, pts.data_ptr(), npts.data_ptr(), Pb.data_ptr(), 0, pts_im.data_ptr(),
                                  n_valid.data_ptr(), depth.data_ptr(), ws.data_ptr(), nb, st) == 0

    ms = timed(run, args.steps, args.warmup)
    kept = int(n_valid.sum())
    nbytes = B * n * 16 + kept * 24 + B * H * W * 8   # scan in, kept points out, depth map out
    o = line("lidar_project", "scans/s", B, ms, args,
             f"{B} scans x {n} points -> {H}x{W} sparse depth (float64), {kept // B} points kept per scan",
             "lidar_project_kernel", nbytes)
    o["dtype"] = "f64"
    if not args.no_cpu_baseline:
        from oracle.lidar import project_scan
        per, k = cpu_time(lambda: project_scan(scan, P, (H, W)), max_n=20)
        o["cpu_baseline"] = {"value": 1 / per, "unit": "scans/s", "cores": 1, "kind": "port",
                             "sample": f"{k} projections of one scan"}
    print(json.dumps(o), flush=True)


def bench_cost_volume(args):
    from depth_estimation_amd._lib import lib
    B, F, D = 2, 32, 64
    h, w = args.height // 2, args.width // 2
    maps = [torch.randn(B, F, h, w, device="cuda") for _ in range(4)]
    left = torch.empty(B, D, 4 * F, h, w, device="cuda")
    right = torch.empty_like(left)
    gmaps = [torch.empty_like(m) for m in maps]
    st = torch.cuda.current_stream().cuda_stream
    L = lib()

    def fwd():
        assert L.dw_cost_volume_fwd(B, F, h, w, D, *(m.data_ptr() for m in maps), left.data_ptr(),
                                    right.data_ptr(), st) == 0

    def bwd():  # the volumes just written stand in for their gradients
        assert L.dw_cost_volume_bwd(B, F, h, w, D, left.data_ptr(), right.data_ptr(),
                                    *(g.data_ptr() for g in gmaps), st) == 0

    nbytes = 2 * left.numel() * 4 + 4 * maps[0].numel() * 4
    wl = f"cost volume of four ({B},{F},{h},{w}) maps, D = {D}"
    for name, fn, kern in (("cost_volume_fwd", fwd, "cost_volume_fwd_kernel"),
                           ("cost_volume_bwd", bwd, "cost_volume_bwd_kernel")):
        ms = timed(fn, args.steps, args.warmup)
        out = line(name, "volume pairs/s", B, ms, args, wl, kern, nbytes)
        if name == "cost_volume_fwd" and not args.no_cpu_baseline:
            from oracle.cost_volume import cost_volume as ref
            cm = [m[:1].cpu() for m in maps]
            per, n = cpu_time(lambda: ref(*cm, maxdisp=D), max_n=5)
            out["cpu_baseline"] = {"value": 1 / per, "unit": "volume pairs/s", "cores": torch.get_num_threads(),
                                   "kind": "port", "sample": f"{n} constructions for one sample"}
        print(json.dumps(out), flush=True)


def bench_soft_argmin(args):
    from depth_estimation_amd._lib import lib
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
