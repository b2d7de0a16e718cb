"""Drop-in modules of the view-synthesis stack over the HIP kernels.

Call signatures and tensor layouts are the usual ones of the public
self-supervised formulation (Godard et al., ICCV 2019), which is what
`BASELINE.json:north_star` names; the reference itself
(/root/reference/model/model.py) holds no such modules.

    disp_to_depth(disp, min_depth, max_depth) -> (scaled_disp, depth)
    BackprojectDepth(B, H, W)(depth, inv_K)   -> (B,4,H*W)
    Project3D(B, H, W, eps)(points, K, T)     -> (B,H,W,2)
    grid_sample_border(img, grid)             -> (B,C,Ho,Wo)   bilinear, border, corners aligned
    SSIM()(x, y)                              -> (B,C,H,W)     clamp((1 - SSIM)/2, 0, 1)
    get_smooth_loss(disp, img)                -> scalar
    PhotometricStack()(...)                   -> dict          the fused fast path

Every op is a torch custom op with autograd, runs on ROCm devices only and
raises when the library is missing (no eager fallback).  Gradients flow to
disparities, depths, points, grids, SSIM inputs and poses; colour images,
intrinsics and the warp source carry none (as on the training path).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import _lib, ops

__all__ = ["disp_to_depth", "BackprojectDepth", "Project3D", "grid_sample_border", "SSIM",
           "get_smooth_loss", "PhotometricStack"]


def _dev(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise _lib.DepthwarpError(f"{name}: expected a ROCm device tensor, got {t.device} "
                                  "(the HIP kernels have no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32(v: float) -> float:
    return float(np.float32(v))


# ------------------------------------------------------------------ disp -> depth
@torch.library.custom_op("depthwarp::disp_to_depth", mutates_args=())
def _disp_to_depth(disp: Tensor, min_disp: float, disp_range: float) -> Tuple[Tensor, Tensor]:
    disp = _dev(disp, "disp")
    scaled, depth = torch.empty_like(disp), torch.empty_like(disp)
    with torch.cuda.device(disp.device):
        rc = _lib.lib().dw_disp_to_depth_fwd(disp.numel(), disp.data_ptr(), min_disp, disp_range,
                                             scaled.data_ptr(), depth.data_ptr(), _stream(disp))
    _lib.check(rc, "dw_disp_to_depth_fwd")
    return scaled, depth


@_disp_to_depth.register_fake
def _(disp, min_disp, disp_range):
    return torch.empty_like(disp), torch.empty_like(disp)


@torch.library.custom_op("depthwarp::disp_to_depth_bwd", mutates_args=())
def _disp_to_depth_bwd(depth: Tensor, g_scaled: Optional[Tensor], g_depth: Optional[Tensor],
                       disp_range: float) -> Tensor:
    out = torch.empty_like(depth)
    gs = None if g_scaled is None else _dev(g_scaled, "g_scaled")
    gd = None if g_depth is None else _dev(g_depth, "g_depth")
    with torch.cuda.device(depth.device):
        rc = _lib.lib().dw_disp_to_depth_bwd(depth.numel(), depth.data_ptr(),
                                             None if gs is None else gs.data_ptr(),
                                             None if gd is None else gd.data_ptr(), disp_range,
                                             out.data_ptr(), _stream(depth))
    _lib.check(rc, "dw_disp_to_depth_bwd")
    return out


@_disp_to_depth_bwd.register_fake
def _(depth, g_scaled, g_depth, disp_range):
    return torch.empty_like(depth)


def _d2d_setup(ctx, inputs, output):
    ctx.range = inputs[2]
    ctx.save_for_backward(output[1])


def _d2d_backward(ctx, g_scaled, g_depth):
    (depth,) = ctx.saved_tensors
    return _disp_to_depth_bwd(depth, g_scaled, g_depth, ctx.range), None, None


_disp_to_depth.register_autograd(_d2d_backward, setup_context=_d2d_setup)


def disp_to_depth(disp: Tensor, min_depth: float, max_depth: float):
    """Sigmoid output -> (scaled disparity, depth); depth in [min_depth, max_depth]."""
    min_disp = 1.0 / max_depth
    max_disp = 1.0 / min_depth
    return _disp_to_depth(disp, _f32(min_disp), _f32(max_disp - min_disp))


# ------------------------------------------------------------------ back-projection
@torch.library.custom_op("depthwarp::backproject", mutates_args=())
def _backproject(depth: Tensor, inv_K: Tensor) -> Tensor:
    depth, inv_K = _dev(depth, "depth"), _dev(inv_K, "inv_K")
    B, _, H, W = depth.shape
    out = torch.empty((B, 4, H * W), dtype=torch.float32, device=depth.device)
    with torch.cuda.device(depth.device):
        rc = _lib.lib().dw_backproject_fwd(B, H, W, depth.data_ptr(), inv_K.data_ptr(),
                                           out.data_ptr(), _stream(depth))
    _lib.check(rc, "dw_backproject_fwd")
    return out


@_backproject.register_fake
def _(depth, inv_K):
    B, _, H, W = depth.shape
    return torch.empty((B, 4, H * W), dtype=torch.float32, device=depth.device)


@torch.library.custom_op("depthwarp::backproject_bwd", mutates_args=())
def _backproject_bwd(inv_K: Tensor, g_points: Tensor, height: int, width: int) -> Tensor:
    g_points = _dev(g_points, "g_points")
    B = g_points.shape[0]
    out = torch.empty((B, 1, height, width), dtype=torch.float32, device=g_points.device)
    with torch.cuda.device(g_points.device):
        rc = _lib.lib().dw_backproject_bwd(B, height, width, inv_K.data_ptr(), g_points.data_ptr(),
                                           out.data_ptr(), _stream(g_points))
    _lib.check(rc, "dw_backproject_bwd")
    return out


@_backproject_bwd.register_fake
def _(inv_K, g_points, height, width):
    return torch.empty((g_points.shape[0], 1, height, width), dtype=torch.float32,
                       device=g_points.device)


def _bp_setup(ctx, inputs, output):
    depth, inv_K = inputs
    ctx.hw = tuple(depth.shape[-2:])
    ctx.save_for_backward(inv_K.contiguous())


def _bp_backward(ctx, g):
    (inv_K,) = ctx.saved_tensors
    return _backproject_bwd(inv_K, g, ctx.hw[0], ctx.hw[1]), None


_backproject.register_autograd(_bp_backward, setup_context=_bp_setup)


class BackprojectDepth(nn.Module):
    """Depth image -> homogeneous camera-space point cloud (B,4,H*W)."""

    def __init__(self, batch_size: int, height: int, width: int):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth: Tensor, inv_K: Tensor) -> Tensor:
        if tuple(depth.shape) != (self.batch_size, 1, self.height, self.width):
            raise ValueError(f"depth: expected {(self.batch_size, 1, self.height, self.width)}, "
                             f"got {tuple(depth.shape)}")
        return _backproject(depth, inv_K)


# ------------------------------------------------------------------ projection
@torch.library.custom_op("depthwarp::project3d", mutates_args=())
def _project3d(points: Tensor, K: Tensor, T: Tensor, height: int, width: int, eps: float) -> Tensor:
    points, K, T = _dev(points, "points"), _dev(K, "K"), _dev(T, "T")
    B = points.shape[0]
    out = torch.empty((B, height, width, 2), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        rc = _lib.lib().dw_project3d_fwd(B, height, width, points.data_ptr(), K.data_ptr(),
                                         T.data_ptr(), eps, out.data_ptr(), _stream(points))
    _lib.check(rc, "dw_project3d_fwd")
    return out


@_project3d.register_fake
def _(points, K, T, height, width, eps):
    return torch.empty((points.shape[0], height, width, 2), dtype=torch.float32, device=points.device)


@torch.library.custom_op("depthwarp::project3d_bwd", mutates_args=())
def _project3d_bwd(points: Tensor, K: Tensor, T: Tensor, g_grid: Tensor, eps: float
                   ) -> Tuple[Tensor, Tensor]:
    g_grid = _dev(g_grid, "g_grid")
    B, H, W, _ = g_grid.shape
    lib = _lib.lib()
    n = lib.dw_project3d_workspace_bytes(B, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device=points.device)
    g_points = torch.empty_like(points)
    g_T = torch.empty((B, 4, 4), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        rc = lib.dw_project3d_bwd(B, H, W, points.data_ptr(), K.data_ptr(), T.data_ptr(), eps,
                                  g_grid.data_ptr(), g_points.data_ptr(), g_T.data_ptr(),
                                  ws.data_ptr(), n, _stream(points))
    _lib.check(rc, "dw_project3d_bwd")
    return g_points, g_T


@_project3d_bwd.register_fake
def _(points, K, T, g_grid, eps):
    return torch.empty_like(points), torch.empty((points.shape[0], 4, 4), dtype=torch.float32,
                                                 device=points.device)


def _p3_setup(ctx, inputs, output):
    points, K, T, h, w, eps = inputs
    ctx.eps = eps
    ctx.save_for_backward(points.contiguous(), K.contiguous(), T.contiguous())


def _p3_backward(ctx, g):
    points, K, T = ctx.saved_tensors
    g_points, g_T = _project3d_bwd(points, K, T, g, ctx.eps)
    return g_points, None, g_T, None, None, None


_project3d.register_autograd(_p3_backward, setup_context=_p3_setup)


class Project3D(nn.Module):
    """Camera-space points -> normalised sampling grid (B,H,W,2) in the view K @ T."""

    def __init__(self, batch_size: int, height: int, width: int, eps: float = 1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points: Tensor, K: Tensor, T: Tensor) -> Tensor:
        if tuple(points.shape) != (self.batch_size, 4, self.height * self.width):
            raise ValueError(f"points: expected {(self.batch_size, 4, self.height * self.width)}, "
                             f"got {tuple(points.shape)}")
        return _project3d(points, K, T, self.height, self.width, _f32(self.eps))


# ------------------------------------------------------------------ bilinear warp
@torch.library.custom_op("depthwarp::warp", mutates_args=())
def _warp(img: Tensor, grid: Tensor) -> Tensor:
    img, grid = _dev(img, "img"), _dev(grid, "grid")
    B, C, H, W = img.shape
    _, Ho, Wo, two = grid.shape
    if two != 2 or grid.shape[0] != B:
        raise ValueError("grid: expected (B,Ho,Wo,2)")
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        rc = _lib.lib().dw_warp_fwd(B, C, H, W, Ho, Wo, img.data_ptr(), grid.data_ptr(),
                                    out.data_ptr(), _stream(img))
    _lib.check(rc, "dw_warp_fwd")
    return out


@_warp.register_fake
def _(img, grid):
    return torch.empty((img.shape[0], img.shape[1], grid.shape[1], grid.shape[2]),
                       dtype=torch.float32, device=img.device)


@torch.library.custom_op("depthwarp::warp_bwd", mutates_args=())
def _warp_bwd(img: Tensor, grid: Tensor, g_out: Tensor) -> Tensor:
    g_out = _dev(g_out, "g_out")
    B, C, H, W = img.shape
    _, Ho, Wo, _ = grid.shape
    g_grid = torch.empty_like(grid)
    with torch.cuda.device(img.device):
        rc = _lib.lib().dw_warp_bwd(B, C, H, W, Ho, Wo, img.data_ptr(), grid.data_ptr(),
                                    g_out.data_ptr(), g_grid.data_ptr(), _stream(img))
    _lib.check(rc, "dw_warp_bwd")
    return g_grid


@_warp_bwd.register_fake
def _(img, grid, g_out):
    return torch.empty_like(grid)


def _w_setup(ctx, inputs, output):
    img, grid = inputs
    ctx.save_for_backward(img.contiguous(), grid.contiguous())


def _w_backward(ctx, g):
    img, grid = ctx.saved_tensors
    return None, _warp_bwd(img, grid, g)


_warp.register_autograd(_w_backward, setup_context=_w_setup)


def grid_sample_border(img: Tensor, grid: Tensor) -> Tensor:
    """`F.grid_sample(img, grid, mode='bilinear', padding_mode='border',
    align_corners=True)`; differentiable w.r.t. `grid` only."""
    if img.requires_grad:
        raise NotImplementedError("grid_sample_border: no gradient w.r.t. the sampled image")
    return _warp(img, grid)


# ------------------------------------------------------------------ SSIM
@torch.library.custom_op("depthwarp::ssim", mutates_args=())
def _ssim(x: Tensor, y: Tensor) -> Tensor:
    x, y = _dev(x, "x"), _dev(y, "y")
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("SSIM: x and y must be (B,C,H,W) of the same shape")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.lib().dw_ssim_fwd(B * C, H, W, x.data_ptr(), y.data_ptr(), out.data_ptr(), _stream(x))
    _lib.check(rc, "dw_ssim_fwd")
    return out


@_ssim.register_fake
def _(x, y):
    return torch.empty_like(x)


@torch.library.custom_op("depthwarp::ssim_bwd", mutates_args=())
def _ssim_bwd(x: Tensor, y: Tensor, g_out: Tensor) -> Tuple[Tensor, Tensor]:
    g_out = _dev(g_out, "g_out")
    B, C, H, W = x.shape
    gx, gy = torch.empty_like(x), torch.empty_like(y)
    with torch.cuda.device(x.device):
        rc = _lib.lib().dw_ssim_bwd(B * C, H, W, x.data_ptr(), y.data_ptr(), g_out.data_ptr(),
                                    gx.data_ptr(), gy.data_ptr(), _stream(x))
    _lib.check(rc, "dw_ssim_bwd")
    return gx, gy


@_ssim_bwd.register_fake
def _(x, y, g_out):
    return torch.empty_like(x), torch.empty_like(y)


def _s_setup(ctx, inputs, output):
    x, y = inputs
    ctx.save_for_backward(x.contiguous(), y.contiguous())


def _s_backward(ctx, g):
    x, y = ctx.saved_tensors
    gx, gy = _ssim_bwd(x, y, g)
    return gx, gy


_ssim.register_autograd(_s_backward, setup_context=_s_setup)


class SSIM(nn.Module):
    """Structural-similarity loss map of an image pair: 3x3 mean windows with
    reflection padding, `clamp((1 - SSIM) / 2, 0, 1)` per pixel and channel."""

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        return _ssim(x, y)


# ------------------------------------------------------------------ smoothness
@torch.library.custom_op("depthwarp::smooth_loss", mutates_args=())
def _smooth(disp: Tensor, img: Tensor) -> Tensor:
    disp, img = _dev(disp, "disp"), _dev(img, "img")
    B, one, H, W = disp.shape
    if one != 1 or tuple(img.shape) != (B, 3, H, W):
        raise ValueError("get_smooth_loss: expected disp (B,1,H,W) and img (B,3,H,W)")
    lib = _lib.lib()
    n = lib.dw_smooth_loss_workspace_bytes(B, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device=disp.device)
    out = torch.empty(1, dtype=torch.float32, device=disp.device)
    with torch.cuda.device(disp.device):
        rc = lib.dw_smooth_loss_fwd(B, H, W, disp.data_ptr(), img.data_ptr(), out.data_ptr(),
                                    ws.data_ptr(), n, _stream(disp))
    _lib.check(rc, "dw_smooth_loss_fwd")
    return out


@_smooth.register_fake
def _(disp, img):
    return torch.empty(1, dtype=torch.float32, device=disp.device)


@torch.library.custom_op("depthwarp::smooth_loss_bwd", mutates_args=())
def _smooth_bwd(disp: Tensor, img: Tensor, g_loss: Tensor) -> Tensor:
    g_loss = _dev(g_loss, "g_loss")
    B, _, H, W = disp.shape
    out = torch.empty_like(disp)
    with torch.cuda.device(disp.device):
        rc = _lib.lib().dw_smooth_loss_bwd(B, H, W, disp.data_ptr(), img.data_ptr(),
                                           g_loss.data_ptr(), out.data_ptr(), _stream(disp))
    _lib.check(rc, "dw_smooth_loss_bwd")
    return out


@_smooth_bwd.register_fake
def _(disp, img, g_loss):
    return torch.empty_like(disp)


def _sm_setup(ctx, inputs, output):
    disp, img = inputs
    ctx.save_for_backward(disp.contiguous(), img.contiguous())


def _sm_backward(ctx, g):
    disp, img = ctx.saved_tensors
    return _smooth_bwd(disp, img, g), None


_smooth.register_autograd(_sm_backward, setup_context=_sm_setup)


def get_smooth_loss(disp: Tensor, img: Tensor) -> Tensor:
    """Edge-aware smoothness of a (mean-normalised) disparity map: first
    differences of `disp` weighted by exp(-|first differences of img|)."""
    return _smooth(disp, img)[0]


# ------------------------------------------------------------------ fused stack
class PhotometricStack(nn.Module):
    """The whole loss stack in one fused pass (see `ops.photometric_forward`)."""

    def forward(self, disp, color, color_pyr, K, inv_K, T, tie_noise=None):
        terms, mask = ops.photometric_forward(disp, color, color_pyr, K, inv_K, T, tie_noise)
        S = len(disp)
        return {"loss": terms[0], "photo": terms[1:1 + S].detach(),
                "smooth": terms[1 + S:].detach(), "argmin": mask & 3,
                "automask": (mask & 3) > 1, "valid": torch.stack([(mask >> 2) & 1, (mask >> 3) & 1], 2)}
