"""Soft-argmin disparity regression as one HIP pass.

Replaces the tail of `DepthBlock.forward` in the reference
(/root/reference/model/model.py:454-456 `view` + `softmax(-out, 1)`, then
:417-429 squeeze, a second `softmax(-prob, 0)`, multiplication with a
materialised disparity grid and the sum): `soft_argmin(out)` takes the
de-convolution output viewed as (B, D, H, W) and returns the (B, H, W)
disparity: at batch 1 that is the reference's (1, H, W) map (model.py:428, a sum over
dim 0 with keepdim after the squeeze), which its callers hand on to 3-D consumers
such as `to_pil_image` (model.py:235).  Differentiable
(the backward recomputes the forward from the cost volume; nothing is saved).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

__all__ = ["soft_argmin"]


def _check(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise _lib.DepthwarpError(f"{name}: expected a ROCm device tensor, got {t.device} "
                                  "(the HIP kernels have no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


@torch.library.custom_op("depthwarp::soft_argmin", mutates_args=())
def _soft_argmin(cost: Tensor) -> Tensor:
    cost = _check(cost, "cost")
    if cost.dim() != 4:
        raise ValueError(f"cost: expected (B, D, H, W), got {tuple(cost.shape)}")
    B, D, H, W = cost.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=cost.device)
    with torch.cuda.device(cost.device):
        rc = _lib.lib().dw_soft_argmin_fwd(B, D, H * W, cost.data_ptr(), out.data_ptr(),
                                           torch.cuda.current_stream(cost.device).cuda_stream)
    _lib.check(rc, "dw_soft_argmin_fwd")
    return out


@_soft_argmin.register_fake
def _(cost):
    B, D, H, W = cost.shape
    return torch.empty((B, 1, H, W), dtype=torch.float32, device=cost.device)


@torch.library.custom_op("depthwarp::soft_argmin_bwd", mutates_args=())
def _soft_argmin_bwd(cost: Tensor, g_disp: Tensor) -> Tensor:
    g_disp = _check(g_disp, "g_disp")
    B, D, H, W = cost.shape
    g_cost = torch.empty_like(cost)
    with torch.cuda.device(cost.device):
        rc = _lib.lib().dw_soft_argmin_bwd(B, D, H * W, cost.data_ptr(), g_disp.data_ptr(),
                                           g_cost.data_ptr(),
                                           torch.cuda.current_stream(cost.device).cuda_stream)
    _lib.check(rc, "dw_soft_argmin_bwd")
    return g_cost


@_soft_argmin_bwd.register_fake
def _(cost, g_disp):
    return torch.empty_like(cost)


def _setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0].contiguous())


def _backward(ctx, g):
    (cost,) = ctx.saved_tensors
    return _soft_argmin_bwd(cost, g)


_soft_argmin.register_autograd(_backward, setup_context=_setup)


def soft_argmin(cost: Tensor) -> Tensor:
    """cost (B, D, H, W), D <= 256 -> disparity (B, H, W) in [0, D-1]."""
    return _soft_argmin(cost)[:, 0]
