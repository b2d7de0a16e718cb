"""Cost-volume construction as one HIP pass per direction.

Replaces `DepthBlock.cost_volume` of the reference
(/root/reference/model/model.py:460-508): same arguments, same two results,
same channel order (including the different order at disparity 0), for any
batch size and feature-map size instead of `Config.batch_size` and
`Config.height // 2` (model.py:478).  Differentiable w.r.t. all four maps.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import _lib

__all__ = ["cost_volume"]


def _check(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise _lib.DepthwarpError(f"{name}: expected a ROCm device tensor, got {t.device} "
                                  "(the HIP kernels have no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


@torch.library.custom_op("depthwarp::cost_volume", mutates_args=())
def _cost_volume(imgl: Tensor, imgr: Tensor, carl: Tensor, carr: Tensor,
                 maxdisp: int) -> Tuple[Tensor, Tensor]:
    maps = [_check(t, n) for t, n in zip((imgl, imgr, carl, carr), ("imgl", "imgr", "carl", "carr"))]
    if maps[0].dim() != 4:
        raise ValueError(f"imgl: expected (B, F, h, w), got {tuple(maps[0].shape)}")
    for t in maps[1:]:
        if t.shape != maps[0].shape:
            # the reference fails in torch.cat with a RuntimeError on a shape mismatch
            raise RuntimeError(f"cost_volume: feature maps differ in shape: "
                               f"{[tuple(m.shape) for m in maps]}")
    B, F, h, w = maps[0].shape
    left = torch.empty((B, maxdisp, 4 * F, h, w), dtype=torch.float32, device=maps[0].device)
    right = torch.empty_like(left)
    with torch.cuda.device(left.device):
        rc = _lib.lib().dw_cost_volume_fwd(B, F, h, w, maxdisp, *(m.data_ptr() for m in maps),
                                           left.data_ptr(), right.data_ptr(), _stream(left))
    _lib.check(rc, "dw_cost_volume_fwd")
    return left, right


@_cost_volume.register_fake
def _(imgl, imgr, carl, carr, maxdisp):
    B, F, h, w = imgl.shape
    left = torch.empty((B, maxdisp, 4 * F, h, w), dtype=torch.float32, device=imgl.device)
    return left, torch.empty_like(left)


@torch.library.custom_op("depthwarp::cost_volume_bwd", mutates_args=())
def _cost_volume_bwd(g_left: Tensor, g_right: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    g_left, g_right = _check(g_left, "g_left"), _check(g_right, "g_right")
    B, D, F4, h, w = g_left.shape
    F = F4 // 4
    outs = [torch.empty((B, F, h, w), dtype=torch.float32, device=g_left.device) for _ in range(4)]
    with torch.cuda.device(g_left.device):
        rc = _lib.lib().dw_cost_volume_bwd(B, F, h, w, D, g_left.data_ptr(), g_right.data_ptr(),
                                           *(o.data_ptr() for o in outs), _stream(g_left))
    _lib.check(rc, "dw_cost_volume_bwd")
    return outs[0], outs[1], outs[2], outs[3]


@_cost_volume_bwd.register_fake
def _(g_left, g_right):
    B, D, F4, h, w = g_left.shape
    return tuple(torch.empty((B, F4 // 4, h, w), dtype=torch.float32, device=g_left.device)
                 for _ in range(4))


def _backward(ctx, g_left, g_right):
    if g_left is None:
        g_left = torch.zeros_like(g_right)
    if g_right is None:
        g_right = torch.zeros_like(g_left)
    gl, gr, gcl, gcr = _cost_volume_bwd(g_left, g_right)
    return gl, gr, gcl, gcr, None


_cost_volume.register_autograd(_backward)


def cost_volume(imgl: Tensor, imgr: Tensor, carl: Tensor, carr: Tensor,
                maxdisp: int = 64) -> Tuple[Tensor, Tensor]:
    """Four (B, F, h, w) maps -> (left_cost, right_cost), each (B, maxdisp, 4F, h, w).

    `maxdisp` is the reference's `self.maxdisp` = `Config.maxdisp // 2` = 64
    (model.py:340); the reference's unused `feature_size` argument is taken
    from the inputs.
    """
    return _cost_volume(imgl, imgr, carl, carr, int(maxdisp))
