"""`bench.py --gpus N` without a launcher starts N ranks as a child job (one per GPU, before this
process touches a GPU) and relays rank 0's JSON line.  No GPU needed: the child is replaced."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_gpus_flag_spawns_one_rank_per_gpu_and_relays_the_line(monkeypatch, capsys):
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        line = json.dumps({"metric": "m", "n_gpus": 4, "config": {"global_batch": 48}})
        return types.SimpleNamespace(returncode=0, stdout="some launcher chatter\n" + line + "\n")

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    # torch.cuda must not be initialised on this path
    import torch
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched")))
    rc = bench.main(["--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]   # the same arguments
    out = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 4


def test_a_failing_child_is_reported(monkeypatch):
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "run",
                        lambda cmd, **kw: types.SimpleNamespace(returncode=3, stdout=""))
    assert bench.main(["--gpus", "2"]) == 3


def test_defaults_and_modes():
    import bench
    a = bench.parse([])
    assert (a.gpus, a.mode, a.batch, a.height, a.width) == (1, "op", 12, 192, 640)
    assert bench.parse(["--mode", "dp"]).batch == 16          # the data-parallel shard
    assert bench.parse(["--mode", "train"]).batch == 12
    assert bench.parse(["--mode", "dp", "--batch", "4"]).batch == 4


def test_counter_file_is_tied_to_the_kernel_sources():
    """`roofline.traffic` comes from a committed counter summary; it is only used when the summary
    carries the hash of the sources the library is built from."""
    import bench
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    doc = json.load(open(path))
    assert set(doc) >= {"sources_sha", "workloads"}
    assert len(bench.kernel_sources_sha()) == 16
    for w in doc["workloads"].values():
        assert w["bytes_corrected"] == 1024.0 * (2 * w["FETCH_SIZE_KB"] + w["WRITE_SIZE_KB"])
