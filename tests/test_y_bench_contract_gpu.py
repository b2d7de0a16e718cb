"""The benchmark prints exactly one JSON line with the agreed fields."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    t_start = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2",
                          "--cpu-batch", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1
    assert d["unit"] == "frames/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 3 * 12 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["limits"][0]["bound"] == "hbm" and r["binding"] in ("hbm", "valu_pipe")
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # the other single-GPU workloads ride in the same line: the high-resolution shard, the
    # data-parallel shard, and the reference's own loss function (one view per call; both views in one)
    sec = d["secondary"]
    assert len(sec) == 4
    for e, (what, units) in zip(sec, (("384x1280, batch=8", 24), ("192x640, batch=16", 48), ("one view per call", 1),
                                       ("both views", 2))):
        assert what in e["workload"], e["workload"]
        assert e["ms_per_step"] > 0 and abs(e["frames_per_s"] - units / (e["ms_per_step"] * 1e-3)) <= 1e-6 * e["frames_per_s"]
        r = e["roofline"]
        assert 0 < r["kernel_ms"] and 0 < r["frac"] < 1
        if units > 2:
            assert r["kernel_ms"] <= e["ms_per_step"] * 1.05   # the dominant kernel is a part of its step
        else:
            # the loss function's two launches replayed as a graph against the rate of Python calls: two
            # clocks of some 15 us each, either may be the larger
            assert r["kernel_ms"] <= e["ms_per_step"] * 3
    # the whole run stays a matter of seconds (the contract: well within minutes)
    assert time.time() - t_start < 60


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks():
    """`--gpus 2` with no launcher in the environment: two ranks are started.  On the one-GPU box
    both share device 0 and the collectives run over gloo (rehearsal); the arithmetic of the line --
    world size, global batch, whole-job frames/s -- is what is checked."""
    env = {"DW_BENCH_REHEARSAL": "1"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    d = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], env)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 24
    assert abs(d["value"] - 3 * 24 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    d = _run(["--gpus", "2", "--mode", "dp", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["config"]["mode"] == "dp"
    assert "all-reduce" in d["config"]["collective"]


def test_train_mode_reports_the_share_of_the_loss_op():
    d = _run(["--mode", "train", "--steps", "5", "--warmup", "2", "--net-width", "16"])
    assert d["config"]["mode"] == "train" and d["n_gpus"] == 1 and d["config"]["global_batch"] == 12
    assert 0 < d["loss_op_share"] < 1 and d["loss_op_ms"] > 0


def _run_launched(args):
    """The multi-GPU command form of the benchmark contract, with one rank: RCCL initialisation,
    barrier, all-reduce and teardown run as they do on a node."""
    sys.path.insert(0, ROOT)
    from bench import free_port
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DW_BENCH_REHEARSAL"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", *args]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_one_rank_under_the_launcher_runs_the_collectives_over_rccl():
    d = _run_launched(["--mode", "dp", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 16 and d["allreduce_ms"] > 0
    assert d["op_only_ms"] > 0 and 0 <= d["exposed_allreduce_ms"] <= d["ms_per_step"]
    d = _run_launched(["--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 12
    d = _run_launched(["--mode", "train", "--steps", "3", "--warmup", "1", "--net-width", "16"])
    assert d["n_gpus"] == 1 and d["config"]["mode"] == "train"


def test_secondary_lines_carry_the_same_fields():
    """bench_next.py prints one line per workload with the fields of the contract line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_next.py"), "--only", "compute_loss", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 4
    for d in lines:
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, k
        r = d["roofline"]
        assert r["kernel"] == "compute_loss_strip_kernel" and 0 < r["frac"] < 1
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert lines[0]["python_call_us"] > 0
