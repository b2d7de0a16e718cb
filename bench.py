#!/usr/bin/env python3
"""Headline benchmark: warp + photometric-loss forward/backward throughput.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode op|dp|train]

--mode op (default)  one step = one fused forward (with gradient stash) + one
    backward of the view-synthesis / photometric-loss stack over one synthetic
    batch of 3-frame sequences: B=12 per GPU, 192x640, 2 source frames, 4 scales.
    Inputs are resident in HBM before the timed region.  The stack shards by
    batch and has no exchange of its own: with N ranks every rank runs its own
    shard (weak scaling) and the slowest rank sets the time.
--mode dp            the data-parallel shard of a training job: B=16 per GPU, and
    a step also all-reduces (RCCL) what a trainer exchanges -- one flat float32
    gradient bucket of the networks (3.2 M floats) plus the scalar loss -- after
    the backward pass.
--mode train         a whole training step: depth + pose networks (stock PyTorch
    convolutions), the HIP loss through a pre-allocated plan, backward, the
    gradient all-reduce and Adam, B=12 per GPU; also reports which share of the
    step the loss op takes.

With --gpus N > 1 and no launcher in the environment this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`
as a child (before anything touches the GPU) and relays its JSON line; under a
launcher (RANK / WORLD_SIZE set) it is one rank.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak
# vector pipes: 256 CUs x 4 SIMDs at 2.4 GHz.  How long an instruction holds its SIMD's pipe depends
# on the opcode and its operands (about 2.55 cycles for plain float32 arithmetic on vector registers,
# about 4.5 for most other opcodes and for any opcode with a scalar-register operand, 8.3 for
# transcendentals: tools/micro/valu_rates.hip); tools/valu_pipe_model.py turns the kernel's
# instruction mix into an average number of pipe cycles per vector instruction.
SIMD_CYCLES_PER_S = 1024 * 2.4e9
DP_BUCKET_FLOATS = 3_200_000  # gradient bucket of the depth + pose networks (about 12.8 MB)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["op", "dp", "train"], default="op")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (default 12; 16 in dp mode)")
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                    help="replay a captured HIP graph, issue the launches eagerly, or (auto) probe both "
                         "during warm-up and time the faster one (op mode)")
    ap.add_argument("--no-graph", action="store_true", help="same as --launch eager")
    ap.add_argument("--net-width", type=int, default=64,
                    help="base channel count of the depth / pose networks in train mode (64: about 4 M parameters)")
    ap.add_argument("--amp", action="store_true",
                    help="train mode: run the networks under bfloat16 autocast (the loss op stays float32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="op mode at one GPU: skip the secondary workloads that follow the headline measurement")
    ap.add_argument("--cpu-batch", type=int, default=4, help="sequences in the CPU baseline sample")
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = 16 if args.mode == "dp" else 12
    return args


# ------------------------------------------------------------------ launching N ranks
def launcher_command(n_gpus: int, argv, port: int):
    """The child command for `--gpus N` without a launcher: one rank per GPU."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), *argv]


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """Runs the ranks as a child job and relays rank 0's JSON line.  Nothing in this
    process has touched the GPU (importing torch does not)."""
    cmd = launcher_command(args.gpus, argv, free_port())
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode else (0 if lines else 1)


# ------------------------------------------------------------------ CPU baseline
def cpu_baseline(args):
    """The CPU oracle (a port: the reference holds no implementation of these
    stages) timed on this host, forward + backward, on a bounded sample."""
    from depth_estimation_amd.synth import make_sequence
    from oracle import photometric as ph
    B = args.cpu_batch
    seq = make_sequence(B, args.height, args.width, seed=0)
    cores = torch.get_num_threads()

    def step():
        disp = [d.clone().requires_grad_(True) for d in seq.disp]
        T = seq.T.clone().requires_grad_(True)
        loss = ph.photometric_stack(disp, seq.color, seq.color_pyr, seq.K, seq.inv_K, T)
        loss.backward()

    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if n >= 3 and el > 10.0:  # about ten seconds of CPU work
            break
        if el > 25.0:
            break
    per = el / n
    return {"value": 3 * B / per, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} fwd+bwd steps of B={B} at {args.height}x{args.width}, 2 sources, "
                      f"4 scales, float32, torch CPU threads={cores}"}


# ------------------------------------------------------------------ roofline of the dominant kernel
def kernel_sources_sha() -> str:
    """Hash of the sources the tile kernel is built from; counter files carry the hash of the
    build they were measured on."""
    h = hashlib.sha256()
    for rel in ("depth_estimation_amd/csrc/photometric.hip", "depth_estimation_amd/csrc/dw_common.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def tile_kernel_ms(B, H, W, S, seq, dev, reps):
    """Average duration of the dominant kernel alone: HIP events on the stream it is launched on."""
    from depth_estimation_amd.ops import PhotometricPlan
    tile_plan = PhotometricPlan(B, H, W, S, dev).bind(*seq.args())
    tile_plan.forward(with_grad=True)   # a complete pass leaves the depth maps the tile kernel reads
    tile_plan.tile_kernel_only()
    for _ in range(10):
        tile_plan.forward(with_grad=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        tile_plan.forward(with_grad=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def roofline(args, seq, dev, alg_total):
    """Times the dominant kernel alone (HIP events on its stream) and prices it against the HBM
    peak by its algorithmic bytes -- the roofline this benchmark is specified against.  What binds
    the kernel is the vector pipes: that limit is reported beside it (`limits`, `binding`) from the
    counters of the last profiler pass over the same sources."""
    B, H, W, S = args.batch, args.height, args.width, 4
    k_ms = tile_kernel_ms(B, H, W, S, seq, dev, max(20, min(args.steps, 200)))
    achieved = alg_total / (k_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": "photo_tile_kernel<true>", "achieved": achieved, "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel_ms": k_ms,
           "algorithmic_bytes": alg_total,
           "note": "priced against HBM as specified; the kernel is bound by the vector pipes, see limits / binding"}
    # Counters cannot be collected from inside the run (the profiler wraps the process): they come
    # from the committed summary of the last rocprofv3 --pmc passes, and only if that summary was
    # measured on the sources this library was built from and on this workload.
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    pmc = None
    if os.path.exists(tpath):
        try:
            pmc = json.load(open(tpath))
        except ValueError:
            pmc = None
    limits = [{"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "frac": achieved / HBM_PEAK_GBPS}]
    if pmc is not None:
        entry = pmc.get("workloads", {}).get(f"{B}x{H}x{W}")
        fresh = pmc.get("sources_sha") == kernel_sources_sha()
        if entry is not None and fresh:
            out["traffic"] = entry["bytes_corrected"]
            insts, per = entry.get("SQ_INSTS_VALU"), pmc.get("valu_pipe_cycles_per_inst")
            if insts and per:
                # time the vector pipes alone need for the instructions of one launch / measured time
                pipe_ms = 1e3 * insts * per / SIMD_CYCLES_PER_S
                limits.append({"bound": "valu_pipe", "achieved": pipe_ms, "peak": k_ms, "unit": "ms of vector-pipe "
                               "time per launch (achieved) within the launch's duration (peak)",
                               "frac": pipe_ms / k_ms, "wave_instructions_per_launch": insts,
                               "pipe_cycles_per_instruction": per,
                               # the same from the counter alone, without the per-opcode cost model:
                               # the fraction if every instruction held its pipe for two cycles, and
                               # the SIMD cycles the launch actually spent per vector instruction
                               "valu_ideal": insts * 2.0 / (SIMD_CYCLES_PER_S * k_ms * 1e-3),
                               "cycles_per_valu_inst": SIMD_CYCLES_PER_S * k_ms * 1e-3 / insts})
        elif entry is not None:
            out["traffic_stale"] = True  # counters of an older build: not reported
    out["limits"] = limits
    out["binding"] = max(limits, key=lambda l: l["frac"])["bound"]
    return out


# ------------------------------------------------------------------ the modes
def run_op(args, ctx):
    """Forward (with gradient stash) + backward of the fused stack; dp mode adds the all-reduce."""
    from depth_estimation_amd.synth import make_sequence, algorithmic_bytes
    from depth_estimation_amd.ops import PhotometricPlan
    dev, dist, rank, world = ctx["dev"], ctx["dist"], ctx["rank"], ctx["world"]
    B, H, W, S = args.batch, args.height, args.width, 4
    seq = make_sequence(B, H, W, S=S, seed=rank).to(dev)
    plan = PhotometricPlan(B, H, W, S, dev).bind(*seq.args())
    dp = args.mode == "dp"
    bucket = None
    if dp:
        # what a data-parallel trainer exchanges per step: the networks' flat gradient bucket with
        # the scalar loss in slot 0 (depth_estimation_amd/train.py::FlatGradBucket)
        bucket = torch.zeros(DP_BUCKET_FLOATS + 1, dtype=torch.float32, device="cpu" if ctx["rehearsal"] else dev)

    def op_step():
        plan.forward(with_grad=True)
        plan.backward()

    def dp_step():
        # As in the trainer, whose gradient hooks start a chunk's all-reduce while backward is still
        # running: the bucket's exchange (the gradients of the networks; here a stand-in of the same
        # size) is started asynchronously before this rank's backward pass is enqueued and waited
        # for after it, the loss (slot 0) follows in an exchange of its own.
        plan.forward(with_grad=True)
        handle = None
        if dist is not None:
            handle = dist.all_reduce(bucket[1:], op=dist.ReduceOp.SUM, async_op=True)
        plan.backward()
        if ctx["rehearsal"]:
            bucket[0] = float(plan.loss[0])
        else:
            bucket[:1].copy_(plan.loss[:1])
        if dist is not None:
            dist.all_reduce(bucket[:1], op=dist.ReduceOp.SUM)
            handle.wait()
        bucket.mul_(1.0 / world)

    step = dp_step if dp else op_step
    for _ in range(3):
        step()
    torch.cuda.synchronize()

    mode = "eager" if (args.no_graph or dp) else args.launch
    graph = None
    if mode != "eager":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
    if mode == "auto":
        # untimed probe: a step is four launches, and which of the two ways of issuing them is
        # faster depends on the host; every rank must take the same decision
        def probe(fn, n=30):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return time.perf_counter() - t
        t_graph, t_eager = probe(graph.replay), probe(step)
        if dist is not None:
            v = torch.tensor([t_graph, t_eager], dtype=torch.float64, device="cpu" if ctx["rehearsal"] else dev)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            t_graph, t_eager = float(v[0]), float(v[1])
        mode = "graph" if t_graph <= t_eager else "eager"
    if mode == "eager":
        graph = None
    run = step if graph is None else graph.replay

    elapsed = timed(run, args, ctx)
    alg = algorithmic_bytes(B, H, W, S)
    per_step = elapsed / args.steps
    what = ("data-parallel shard: warp+SSIM/L1+smoothness fwd/bwd + all-reduce of a "
            f"{DP_BUCKET_FLOATS}-float gradient bucket and the loss" if dp else
            "warp+SSIM/L1+smoothness fwd/bwd microbench")
    out = {
        "metric": "warp+photometric-loss fwd+bwd throughput",
        "value": 3 * B * world / per_step, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{what}, {H}x{W}, batch={B}/GPU, 2 source frames, 4 scales",
                   "mode": args.mode, "global_batch": B * world,
                   "launch": "eager" if graph is None else "hip-graph",
                   "collective": ("all-reduce (sum) of %d floats per step, started before this rank's backward "
                                  "pass is enqueued and waited for after it, %s" %
                                  (DP_BUCKET_FLOATS + 1, "gloo rehearsal" if ctx["rehearsal"] else "RCCL")) if dp
                   else "none: the stack shards by batch and exchanges nothing"},
        "hbm_gbps_algorithmic": world * alg["total"] / per_step / 1e9,
        "loss": float(plan.loss[0].item()),
    }
    if dp and dist is not None and not ctx["rehearsal"]:
        # the collective alone, same payload, for the record
        for _ in range(5):
            dist.all_reduce(bucket)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(bucket)
        torch.cuda.synchronize()
        out["allreduce_ms"] = 1e3 * (time.perf_counter() - t) / 20
        # the same step without the exchange: what of the all-reduce the step did not hide
        for _ in range(5):
            op_step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            op_step()
        torch.cuda.synchronize()
        out["op_only_ms"] = 1e3 * (time.perf_counter() - t) / 50
        out["exposed_allreduce_ms"] = max(0.0, 1e3 * per_step - out["op_only_ms"])
    if rank == 0:
        out["roofline"] = roofline(args, seq, dev, alg["total"])
        if world == 1 and not dp and not args.no_secondary and (B, H, W) == (12, 192, 640):
            out["secondary"] = secondary_workloads(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    return out


def secondary_workloads(dev, steps=60):
    """The other single-GPU workloads of the benchmark's configurations, timed after the headline
    one in the same run (each `steps` steps after 10 of warm-up, HIP events; eager launches):
    the high-resolution shard (8 x 384x1280), the data-parallel shard (16 x 192x640, with the
    bucket and loss bookkeeping of a data-parallel step but no exchange: one rank), and the
    reference's own loss function at its own sizes -- one view per call and both views in one."""
    from depth_estimation_amd.synth import make_sequence, algorithmic_bytes
    from depth_estimation_amd.ops import PhotometricPlan
    from depth_estimation_amd.loss import compute_loss, compute_loss_lr
    from depth_estimation_amd._lib import lib

    def events(fn, n, warm=10):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    res = []
    for name, B, H, W, bucketed in (("high-resolution shard", 8, 384, 1280, False),
                                    ("data-parallel shard at one rank", 16, 192, 640, True)):
        seq = make_sequence(B, H, W, S=4, seed=0).to(dev)
        plan = PhotometricPlan(B, H, W, 4, dev).bind(*seq.args())
        bucket = torch.zeros(DP_BUCKET_FLOATS + 1, dtype=torch.float32, device=dev) if bucketed else None

        def step():
            plan.forward(with_grad=True)
            plan.backward()
            if bucket is not None:
                bucket[:1].copy_(plan.loss[:1])
                bucket.mul_(1.0)

        ms = events(step, steps)
        alg = algorithmic_bytes(B, H, W, 4)["total"]
        k_ms = tile_kernel_ms(B, H, W, 4, seq, dev, steps)
        res.append({"workload": f"{name}: warp+SSIM/L1+smoothness fwd/bwd, {H}x{W}, batch={B}, 2 source frames, "
                                "4 scales" + (", + bucket bookkeeping of %d floats" % (DP_BUCKET_FLOATS + 1) if bucketed else ""),
                    "ms_per_step": ms, "frames_per_s": 3 * B / (ms * 1e-3),
                    "roofline": {"bound": "hbm", "kernel": "photo_tile_kernel<true>", "kernel_ms": k_ms,
                                 "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": alg}})
        del plan, seq, bucket
    # Train.compute_loss at the reference's sizes (/root/reference/train.py:272-309, config.py:20-21):
    # float64 375x1242 ground truth, 256x512 prediction; train.py:113 evaluates it on two views
    Ht, Wt, H, W = 375, 1242, 256, 512
    g = torch.Generator().manual_seed(0)
    lt, rt = (torch.rand(1, Ht, Wt, generator=g, dtype=torch.float64).to(dev) for _ in range(2))
    lp, rp = (torch.rand(1, H, W, generator=g).to(dev) for _ in range(2))
    L = lib()
    nb = L.dw_compute_loss_workspace_bytes(2, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out9 = torch.empty(9, device=dev)
    F64 = 2  # DW_CL_FLAG_TARGET_F64

    def raw_one():
        assert L.dw_compute_loss_fwd(1, Ht, Wt, H, W, lt.data_ptr(), lp.data_ptr(), out9.data_ptr(), F64,
                                     ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream) == 0

    def raw_pair():
        assert L.dw_compute_loss_pair_fwd(Ht, Wt, H, W, lt.data_ptr(), lp.data_ptr(), rt.data_ptr(), rp.data_ptr(),
                                          out9.data_ptr(), F64, ws.data_ptr(), nb,
                                          torch.cuda.current_stream().cuda_stream) == 0

    def wall(fn, n):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    for name, views, call, raw in (("one view per call", 1, lambda: compute_loss(lt, lp), raw_one),
                                   ("both views of a step in one call (compute_loss_lr)", 2,
                                    lambda: compute_loss_lr(lt, lp, rt, rp), raw_pair)):
        raw()
        # the two launches of a call without the host between them: sixteen calls recorded back to back in
        # one graph (replaying a graph of a single call would time the submission of the graph instead)
        per_graph = 16
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(per_graph):
                raw()
        k_ms = events(graph.replay, max(steps // 4, 10)) / per_graph
        ms = wall(call, max(steps, 200))     # the Python call as the reference would make it, host included
        nbytes = views * (8 * Ht * Wt + 4 * H * W)
        res.append({"workload": f"Train.compute_loss, {name}: float64 ({Ht},{Wt}) ground truth against ({H},{W}) predictions",
                    "ms_per_step": ms, "frames_per_s": views / (ms * 1e-3),
                    "roofline": {"bound": "hbm", "kernel": "compute_loss_strip_kernel + finalisation", "kernel_ms": k_ms,
                                 "frac": nbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": nbytes}})
    return res


def run_train(args, ctx):
    """Whole training steps of the per-GPU loop (depth_estimation_amd/train.py)."""
    from depth_estimation_amd.synth import make_sequence, algorithmic_bytes
    from depth_estimation_amd.train import Trainer
    from depth_estimation_amd.ops import PhotometricPlan
    dev, rank, world = ctx["dev"], ctx["rank"], ctx["world"]
    B, H, W, S = args.batch, args.height, args.width, 4
    seq = make_sequence(B, H, W, S=S, seed=rank).to(dev)
    trainer = Trainer(dev, width=args.net_width, amp=args.amp)
    for _ in range(3):
        trainer.step(seq)
    torch.cuda.synchronize()
    last = {}

    def step():
        last["loss"] = trainer.step(seq)

    elapsed = timed(step, args, ctx)
    per_step = elapsed / args.steps
    # the loss op alone on the same inputs, for its share of the step
    with torch.no_grad():
        disp = [d.detach().contiguous() for d in trainer.depth_net(seq.color[0])]
    plan = PhotometricPlan(B, H, W, S, dev).bind(disp, seq.color, seq.color_pyr, seq.K, seq.inv_K, seq.T)
    for _ in range(5):
        plan.forward(True)
        plan.backward()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        plan.forward(True)
        plan.backward()
    e1.record()
    torch.cuda.synchronize()
    op_ms = e0.elapsed_time(e1) / 50
    n_params = sum(p.numel() for p in trainer.bucket.params)
    alg = algorithmic_bytes(B, H, W, S)
    return {
        "metric": "full training step throughput", "value": 3 * B * world / per_step, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (networks under bf16 autocast)" if args.amp else "f32", "data": "synthetic",
        "config": {"workload": f"full training step (depth + pose networks on PyTorch-ROCm, HIP warp/loss op, "
                               f"gradient all-reduce, Adam), {H}x{W}, batch={B}/GPU, 2 source frames, 4 scales",
                   "mode": "train", "global_batch": B * world, "network_parameters": n_params},
        "loss": float(last["loss"]),
        "loss_op_ms": op_ms, "loss_op_share": op_ms / (1e3 * per_step),
        "hbm_gbps_algorithmic": world * alg["total"] / per_step / 1e9,
    }


def timed(run, args, ctx):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    dist, dev = ctx["dist"], ctx["dev"]
    for _ in range(args.warmup):
        run()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if ctx["rehearsal"] else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a one-GPU box only: DW_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and uses gloo for the collectives (RCCL refuses two ranks per device).
    rehearsal = os.environ.get("DW_BENCH_REHEARSAL") == "1"
    dist = None
    if "WORLD_SIZE" in os.environ:
        # under a launcher the collectives run even at world size 1 (one-rank RCCL communicator),
        # so that a one-GPU box exercises the same calls as a node does
        import torch.distributed as dist
        if rehearsal:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        local = 0
        torch.cuda.set_device(0)
    ctx = {"dev": torch.device("cuda", local), "dist": dist, "rank": rank, "world": world, "rehearsal": rehearsal}
    out = run_train(args, ctx) if args.mode == "train" else run_op(args, ctx)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
