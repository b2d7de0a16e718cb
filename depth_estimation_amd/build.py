"""Builds libdepthwarp.so (the HIP kernels + C ABI) in-tree with hipcc.

    python -m depth_estimation_amd.build [--force]

The library is compiled for gfx950 only.  `-ffp-contract=off` is part of the
numerical contract of the forward kernels (no fused multiply-add where a
validity bit or an arg-min index is decided); backward code asks for fmaf
explicitly.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libdepthwarp.so"
SOURCES = ["photometric.hip", "compute_loss.hip", "modules.hip", "soft_argmin.hip", "cost_volume.hip", "lidar.hip", "abi.hip"]
# per-source options: the tile kernel runs 1 % faster with the scheduler's max-ILP strategy
# (measured), the soft-argmin backward kernel much slower, so the option is not global
EXTRA_FLAGS = {"photometric.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
HEADERS = [CSRC / "dw_common.h", ROOT / "include" / "depthwarp.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES if (CSRC / s).exists()] + HEADERS
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not stale():
        return LIB
    common = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
              "-fPIC", "-fvisibility=hidden", "-DDW_BUILD", f"-I{ROOT / 'include'}", f"-I{CSRC}"]
    objdir = ROOT / "build" / "obj"
    objdir.mkdir(parents=True, exist_ok=True)
    procs = []
    for src in SOURCES:
        if not (CSRC / src).exists():
            continue
        obj = objdir / (Path(src).stem + ".o")
        cmd = [_hipcc(), *common, *EXTRA_FLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    objs = []
    for cmd, obj, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        objs.append(str(obj))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(LIB)]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
