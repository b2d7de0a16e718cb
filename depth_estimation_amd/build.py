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
    srcs = [str(CSRC / s) for s in SOURCES if (CSRC / s).exists()]
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
           "-fPIC", "-shared", "-fvisibility=hidden", "-DDW_BUILD",
           f"-I{ROOT / 'include'}", f"-I{CSRC}", *srcs, "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
