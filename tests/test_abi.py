"""The C-ABI library loads without a GPU, exports every symbol the header
declares, and its host-only entry points behave (no kernels are launched)."""
import ctypes as C
import os
import re

import pytest

from depth_estimation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "depthwarp.h")).read()
    return re.findall(r"DW_API\s+[\w\s\*]+?\b(dw_\w+)\s*\(", text)


def test_header_symbols_are_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 7 and "dw_photometric_fwd" in names
    handle = C.CDLL(str(_lib.LIB_PATH))
    for n in names:
        assert hasattr(handle, n), f"{n} declared in depthwarp.h but not exported"
        assert n in _lib._SIGNATURES, f"{n} has no ctypes signature in _lib.py"


def test_version_and_error_strings():
    lib = _lib.lib()
    assert lib.dw_version() == 2
    assert lib.dw_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert len(lib.dw_strerror(code)) > 3
    assert b"unknown" in lib.dw_strerror(-99)


def test_workspace_sizes_and_shape_checks():
    lib = _lib.lib()
    d = _lib.DwDesc(12, 192, 640, 4, 2, 0)
    n = lib.dw_photometric_workspace_bytes(d)
    # at least the full-resolution disparity gradient and the smoothness adjoint of every scale
    assert n % 256 == 0 and n >= 2 * 12 * 192 * 640 * 4
    assert lib.dw_photometric_workspace_bytes(_lib.DwDesc(1, 100, 640, 4, 2, 0)) == 0  # 100 % 8 != 0
    assert lib.dw_photometric_workspace_bytes(_lib.DwDesc(1, 192, 640, 5, 2, 0)) == 0
    assert lib.dw_photometric_workspace_bytes(_lib.DwDesc(1, 192, 640, 4, 3, 0)) == 0
    # H * W <= 2^28 (32-bit byte offsets inside a colour plane): one row more is refused
    assert lib.dw_photometric_workspace_bytes(_lib.DwDesc(1, 16384, 16384, 4, 2, 0)) > 0
    assert lib.dw_photometric_workspace_bytes(_lib.DwDesc(1, 16384, 16392, 4, 2, 0)) == 0
    assert lib.dw_compute_loss_workspace_bytes(1, 256, 512) > 0
    assert lib.dw_compute_loss_workspace_bytes(1, 2, 512) == 0


def test_null_and_bad_arguments_return_codes_without_launching():
    lib = _lib.lib()
    d = _lib.DwDesc(1, 32, 64, 4, 2, 0)
    rc = lib.dw_photometric_fwd(d, None, None, None, None, None, None, None, None, None, None, 0, 1, None)
    assert rc == -1
    bad = _lib.DwDesc(1, 33, 64, 4, 2, 0)
    rc = lib.dw_photometric_fwd(bad, None, None, None, None, None, None, None, None, None, None, 0, 1, None)
    assert rc == -2
    assert lib.dw_compute_loss_fwd(1, 8, 8, 8, 8, None, None, None, 0, None, 0, None) == -1
    with pytest.raises(_lib.DepthwarpError):
        _lib.check(-3, "x")


def test_ops_refuse_cpu_tensors():
    import torch
    from depth_estimation_amd import ops
    from depth_estimation_amd.loss import compute_loss
    from depth_estimation_amd.synth import make_sequence
    seq = make_sequence(1, 16, 32, seed=0)
    with pytest.raises(_lib.DepthwarpError):
        ops.photometric_stack(*seq.args())
    with pytest.raises(_lib.DepthwarpError):
        compute_loss(torch.rand(1, 16, 16), torch.rand(1, 16, 16))


def test_flags_in_header_and_binding_agree_and_size_the_workspace():
    text = open(os.path.join(ROOT, "include", "depthwarp.h")).read()
    flags = {k: int(v) for k, v in re.findall(r"#define\s+(DW_FLAG_\w+)\s+(\d+)", text)}
    assert flags == {"DW_FLAG_TILE_KERNEL_ONLY": _lib.DW_FLAG_TILE_KERNEL_ONLY,
                     "DW_FLAG_NO_TAIL_SPLIT": _lib.DW_FLAG_NO_TAIL_SPLIT,
                     "DW_FLAG_COLOR_GRAD": _lib.DW_FLAG_COLOR_GRAD}
    lib = _lib.lib()
    B, H, W = 3, 64, 96
    plain = lib.dw_photometric_workspace_bytes(_lib.DwDesc(B, H, W, 4, 2, 0))
    color = lib.dw_photometric_workspace_bytes(_lib.DwDesc(B, H, W, 4, 2, _lib.DW_FLAG_COLOR_GRAD))
    # at least: the adjoints of the warped colours at every scale (2 frames x 3 channels, float32)
    # and the fixed-point image of the rare path (2 source images, three channels, 64 bit)
    assert color - plain >= 4 * B * 6 * H * W * 4 + 2 * B * 3 * H * W * 8
    assert lib.dw_photometric_workspace_bytes(_lib.DwDesc(B, H, W, 4, 2, 8)) == 0  # unknown flag
    # the colour-gradient read-out refuses a descriptor without its flag, before touching any pointer
    assert lib.dw_photometric_bwd_color(_lib.DwDesc(B, H, W, 4, 2, 0), None, None, 0, None, None) < 0
    assert lib.dw_photometric_bwd_color(_lib.DwDesc(B, H, W, 4, 2, 4), None, None, 0, None, None) == -1
