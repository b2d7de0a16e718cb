"""Evaluation metrics of the reference on the HIP path.

Mirrors /root/reference/train.py: `compute_errors` (:311-336),
`compute_depth_errors` (:338-343) and `normalize_image` (:248-253), and offers
the fused form `compute_loss_and_metrics`, which produces the loss of
`Train.compute_loss` and these metrics from one pass over the same two tensors
(the reference makes three passes and two device-to-host copies).

Batch-generalised like `loss.compute_loss`: per-sample results; a batch of one
returns 0-dim tensors where the reference returns scalars.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib
from .loss import _as_b1hw

__all__ = ["compute_loss_and_metrics", "compute_errors", "compute_depth_errors", "normalize_image"]

MAXDISP = 128.0   # Config.maxdisp, /root/reference/config/config.py:32


def _need_cuda(*ts: Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise _lib.DepthwarpError("expected ROCm device tensors (the HIP kernels have no CPU path)")


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _scalar_if_single(t: Tensor) -> Tensor:
    return t[0] if t.numel() == 1 else t


@torch.library.custom_op("depthwarp::compute_loss_metrics_fwd", mutates_args=())
def _loss_metrics(target: Tensor, pred: Tensor, clamp_min: float, clamp_max: float,
                  antialias: bool = False) -> Tensor:
    """target (B,1,Ht,Wt) float32 or float64 (read in place, rounded to float32 by the kernel),
    pred (B,1,H,W) float32 -> (B,8)."""
    _need_cuda(target, pred)
    if target.dtype not in (torch.float32, torch.float64) or pred.dtype != torch.float32:
        raise TypeError(f"expected a float32 / float64 target and a float32 prediction, got {target.dtype}, {pred.dtype}")
    B, _, H, W = pred.shape
    if target.shape[0] != B:
        raise RuntimeError("target and pred disagree on the batch size")
    target, pred = target.contiguous(), pred.contiguous()
    lib = _lib.lib()
    nbytes = lib.dw_compute_loss_workspace_bytes(B, H, W)
    if nbytes == 0:
        raise _lib.DepthwarpError(f"unsupported shape B={B} H={H} W={W}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
    out = torch.empty((B, 8), dtype=torch.float32, device=pred.device)
    with torch.cuda.device(pred.device):
        rc = lib.dw_compute_loss_metrics_fwd(B, target.shape[2], target.shape[3], H, W, target.data_ptr(),
                                             pred.data_ptr(), clamp_min, clamp_max, out.data_ptr(),
                                             (_lib.DW_CL_FLAG_ANTIALIAS if antialias else 0) |
                                             (_lib.DW_CL_FLAG_TARGET_F64 if target.dtype == torch.float64 else 0),
                                             ws.data_ptr(), nbytes, _stream(pred))
    _lib.check(rc, "dw_compute_loss_metrics_fwd")
    return out


@_loss_metrics.register_fake
def _(target, pred, clamp_min, clamp_max, antialias=False):
    return torch.empty((pred.shape[0], 8), dtype=torch.float32, device=pred.device)


@torch.library.custom_op("depthwarp::compute_loss_metrics_norm_fwd", mutates_args=())
def _loss_metrics_norm(target: Tensor, pred: Tensor, clamp_min: float, clamp_max: float,
                       antialias: bool = False) -> tuple[Tensor, Tensor]:
    """As `_loss_metrics`, plus the prediction normalised per sample (`normalize_image`) from the same
    two launches: -> (B,8), (B,1,H,W)."""
    _need_cuda(target, pred)
    if target.dtype not in (torch.float32, torch.float64) or pred.dtype != torch.float32:
        raise TypeError(f"expected a float32 / float64 target and a float32 prediction, got {target.dtype}, {pred.dtype}")
    B, _, H, W = pred.shape
    if target.shape[0] != B:
        raise RuntimeError("target and pred disagree on the batch size")
    target, pred = target.contiguous(), pred.contiguous()
    lib = _lib.lib()
    nbytes = lib.dw_compute_loss_workspace_bytes(B, H, W)
    if nbytes == 0:
        raise _lib.DepthwarpError(f"unsupported shape B={B} H={H} W={W}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
    out = torch.empty((B, 8), dtype=torch.float32, device=pred.device)
    norm = torch.empty_like(pred)
    with torch.cuda.device(pred.device):
        rc = lib.dw_compute_loss_metrics_norm_fwd(B, target.shape[2], target.shape[3], H, W, target.data_ptr(),
                                                  pred.data_ptr(), clamp_min, clamp_max, out.data_ptr(),
                                                  norm.data_ptr(),
                                                  (_lib.DW_CL_FLAG_ANTIALIAS if antialias else 0) |
                                                  (_lib.DW_CL_FLAG_TARGET_F64 if target.dtype == torch.float64 else 0),
                                                  ws.data_ptr(), nbytes, _stream(pred))
    _lib.check(rc, "dw_compute_loss_metrics_norm_fwd")
    return out, norm


@_loss_metrics_norm.register_fake
def _(target, pred, clamp_min, clamp_max, antialias=False):
    return (torch.empty((pred.shape[0], 8), dtype=torch.float32, device=pred.device), torch.empty_like(pred))


def compute_loss_and_metrics(target: Tensor, pred: Tensor, maxdisp: float = MAXDISP,
                             antialias: bool = False, normalized: bool = False) -> dict:
    """One pass: dict of (B,) tensors loss, ssim_term, l1, smooth (train.py:272-309),
    rmse (train.py:311-343, one side) and pred_min / pred_max (train.py:250-251).
    `normalized=True` adds "pred_normalized" (B,1,H,W) = `normalize_image(pred)` (train.py:248-253),
    written by the launch that finishes the sums: two launches for everything."""
    with torch.no_grad():
        target = _as_b1hw("target", target)
        if target.dtype not in (torch.float32, torch.float64):
            target = target.float()   # a float64 ground truth (the reference's) is read as it is
        p4 = _as_b1hw("pred", pred).float()
        norm = None
        if normalized:
            out, norm = _loss_metrics_norm(target, p4, 1e-3, float(maxdisp), antialias)
        else:
            out = _loss_metrics(target, p4, 1e-3, float(maxdisp), antialias)
    names = ("loss", "ssim_term", "l1", "smooth", "rmse", "pred_min", "pred_max")
    res = {k: out[:, i] for i, k in enumerate(names)}
    if normalized:
        res["pred_normalized"] = norm
    return res


def compute_errors(inputs: dict, outputs: dict, maxdisp: float = MAXDISP, antialias: bool = False):
    """Drop-in for `Train.compute_errors(inputs, outputs)`: (left_rmse, right_rmse).
    The output size is the prediction's (the reference reads Config.height/width)."""
    res = []
    for side in ("l", "r"):
        m = compute_loss_and_metrics(inputs[f"{side}_depth"], outputs[f"{side}_pred"], maxdisp, antialias)
        res.append(_scalar_if_single(m["rmse"]))
    return res[0], res[1]


def compute_depth_errors(gt: Tensor, pred: Tensor) -> Tensor:
    """Drop-in for `Train.compute_depth_errors(gt, pred)`: sqrt(mean((gt - pred)^2)),
    per sample along dim 0 (a batch of one gives the reference's scalar)."""
    _need_cuda(gt, pred)
    if gt.shape != pred.shape:
        gt, pred = torch.broadcast_tensors(gt, pred)
    gt, pred = gt.float().contiguous(), pred.float().contiguous()
    B = gt.shape[0]
    n = gt.numel() // B
    lib = _lib.lib()
    nbytes = lib.dw_normalize_image_workspace_bytes(B, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gt.device)
    out = torch.empty(B, dtype=torch.float32, device=gt.device)
    with torch.cuda.device(gt.device):
        rc = lib.dw_rmse(B, n, gt.data_ptr(), pred.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes,
                         _stream(gt))
    _lib.check(rc, "dw_rmse")
    return _scalar_if_single(out)


def normalize_image(x: Tensor, minmax: Tensor = None) -> Tensor:
    """Drop-in for `Train.normalize_image(x)`, per sample along dim 0 and without
    host round trips.  `minmax` (B, 2) may carry known extrema, e.g.
    `torch.stack([m["pred_min"], m["pred_max"]], 1)` of `compute_loss_and_metrics`."""
    _need_cuda(x)
    x = x.float().contiguous()
    B = x.shape[0]
    n = x.numel() // B
    lib = _lib.lib()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        if minmax is not None:
            minmax = minmax.float().contiguous()
            if tuple(minmax.shape) != (B, 2):
                raise ValueError(f"minmax: expected ({B}, 2), got {tuple(minmax.shape)}")
            rc = lib.dw_normalize_image(B, n, x.data_ptr(), minmax.data_ptr(), 2, out.data_ptr(), None, 0,
                                        _stream(x))
        else:
            nbytes = lib.dw_normalize_image_workspace_bytes(B, n)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            rc = lib.dw_normalize_image(B, n, x.data_ptr(), None, 0, out.data_ptr(), ws.data_ptr(), nbytes,
                                        _stream(x))
    _lib.check(rc, "dw_normalize_image")
    return out
