"""ctypes binding of libdepthwarp.so (declarations: include/depthwarp.h).

There is no fallback: if the library has not been built, or an entry point
returns an error code, the call raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

_PKG = Path(__file__).resolve().parent
# DEPTHWARP_LIB selects another build of the same library (kernel tuning A/B runs).
LIB_PATH = Path(os.environ.get("DEPTHWARP_LIB", _PKG / "libdepthwarp.so"))

DW_MAX_SCALES = 4
DW_NUM_SRC = 2
DW_FLAG_TILE_KERNEL_ONLY = 1
DW_FLAG_NO_TAIL_SPLIT = 2
DW_FLAG_COLOR_GRAD = 4
DW_CL_FLAG_ANTIALIAS = 1
DW_CL_FLAG_TARGET_F64 = 2
DW_STATIC_SCRATCH_BYTES = 1 << 20


class DwDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("S", C.c_int32), ("F", C.c_int32), ("flags", C.c_int32)]


class DepthwarpError(RuntimeError):
    pass


_lib = None

_VP = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_SIGNATURES = {
    "dw_version": (C.c_int, []),
    "dw_strerror": (C.c_char_p, [C.c_int]),
    "dw_photometric_workspace_bytes": (C.c_size_t, [C.POINTER(DwDesc)]),
    "dw_photometric_fwd": (C.c_int, [C.POINTER(DwDesc), _PP, _PP, _PP, _VP, _VP, _VP, _VP, _PP,
                                     _VP, _VP, C.c_size_t, C.c_int, _VP]),
    "dw_photometric_bwd": (C.c_int, [C.POINTER(DwDesc), _PP, _VP, _VP, C.c_int32, _VP, C.c_size_t, _PP,
                                     _VP, _VP]),
    "dw_photometric_bwd_color": (C.c_int, [C.POINTER(DwDesc), _VP, _VP, C.c_size_t, _PP, _VP]),
    "dw_compute_loss_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "dw_compute_loss_fwd": (C.c_int, [C.c_int32] * 5 + [_VP, _VP, _VP, C.c_int32, _VP, C.c_size_t, _VP]),
    "dw_compute_loss_pair_fwd": (C.c_int, [C.c_int32] * 4 + [_VP, _VP, _VP, _VP, _VP, C.c_int32, _VP, C.c_size_t, _VP]),
    "dw_compute_loss_fwd_static": (C.c_int, [C.c_int32] * 5 + [_VP, _VP, _VP, C.c_int32, _VP]),
    "dw_compute_loss_metrics_fwd": (C.c_int, [C.c_int32] * 5 + [_VP, _VP, C.c_float, C.c_float, _VP, C.c_int32,
                                                            _VP, C.c_size_t, _VP]),
    "dw_compute_loss_metrics_norm_fwd": (C.c_int, [C.c_int32] * 5 + [_VP, _VP, C.c_float, C.c_float, _VP, _VP, C.c_int32,
                                                   _VP, C.c_size_t, _VP]),
    "dw_normalize_image_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int64]),
    "dw_normalize_image": (C.c_int, [C.c_int32, C.c_int64, _VP, _VP, C.c_int32, _VP, _VP, C.c_size_t, _VP]),
    "dw_rmse": (C.c_int, [C.c_int32, C.c_int64, _VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    "dw_disp_to_depth_fwd": (C.c_int, [C.c_int64, _VP, C.c_float, C.c_float, _VP, _VP, _VP]),
    "dw_disp_to_depth_bwd": (C.c_int, [C.c_int64, _VP, _VP, _VP, C.c_float, _VP, _VP]),
    "dw_backproject_fwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, _VP]),
    "dw_backproject_bwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, _VP]),
    "dw_project3d_fwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, C.c_float, _VP, _VP]),
    "dw_project3d_workspace_bytes": (C.c_size_t, [C.c_int32] * 3),
    "dw_project3d_bwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _VP,
                                                      C.c_size_t, _VP]),
    "dw_warp_fwd": (C.c_int, [C.c_int32] * 6 + [_VP, _VP, _VP, _VP]),
    "dw_warp_bwd": (C.c_int, [C.c_int32] * 6 + [_VP, _VP, _VP, _VP, _VP]),
    "dw_ssim_fwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, _VP]),
    "dw_ssim_bwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, _VP, _VP, _VP]),
    "dw_smooth_loss_workspace_bytes": (C.c_size_t, [C.c_int32] * 3),
    "dw_smooth_loss_fwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    "dw_smooth_loss_bwd": (C.c_int, [C.c_int32] * 3 + [_VP, _VP, _VP, _VP, _VP]),
    "dw_soft_argmin_fwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _VP, _VP, _VP]),
    "dw_soft_argmin_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _VP, _VP, _VP, _VP]),
    "dw_lidar_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "dw_lidar_project": (C.c_int, [C.c_int32] * 4 + [_VP, _VP, _VP, C.c_int32, _VP, _VP, _VP, _VP, C.c_size_t,
                                                      _VP]),
    "dw_cost_volume_fwd": (C.c_int, [C.c_int32] * 5 + [_VP] * 7),
    "dw_cost_volume_bwd": (C.c_int, [C.c_int32] * 5 + [_VP] * 7),
}


def lib():
    """The loaded library. Raises DepthwarpError if it is missing."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise DepthwarpError(
                f"{LIB_PATH} is missing: build it with `python -m depth_estimation_amd.build` "
                "(hipcc, gfx950). There is no CPU or PyTorch fallback for these ops.")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().dw_strerror(code).decode()
        raise DepthwarpError(f"{what} failed: {msg} (code {code})")


def ptr_array(tensors, n=None):
    n = n if n is not None else len(tensors)
    arr = (C.c_void_p * n)()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr
