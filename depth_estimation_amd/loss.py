"""`compute_loss(target, pred)`: the reference's loss function as one HIP call.

Mirrors `Train.compute_loss(self, target, pred)` at
/root/reference/train.py:272-309 (`self` is unused there): SSIM(7x7) + L1 + a
Sobel term on the resized ground truth, mixed 0.75 / 0.20 / 0.15, computed
without gradient exactly as in the reference (its body runs under `no_grad`,
train.py:274; the caller marks the scalar `requires_grad_` itself, train.py:114).

The reference is hard-wired to batch 1 (`view(1, 1, H, W)`, train.py:277-279).
Here a batch is allowed: sample b of the result equals the reference call on
sample b (to float32 rounding: the grouping of an image's partial sums depends on
the batch size); a batch of one returns a 0-dim tensor like the reference.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

__all__ = ["compute_loss", "compute_loss_lr", "compute_loss_terms", "compute_loss_into"]


def _as_b1hw(name: str, t: Tensor) -> Tensor:
    if t.dim() == 3:
        t = t[:, None]
    if t.dim() != 4 or t.shape[1] != 1:
        raise RuntimeError(f"{name}: expected (B,H,W) or (B,1,H,W), got {tuple(t.shape)}")
    return t


# Partial-sum workspaces, one per (device, stream): a call's kernels run on the current
# stream, so calls that could overlap never share one.  Sized by the library's upper bound
# (32 bytes per sample, strip and pair of rows: 16 KB for one 256 x 512 sample, 1 MB at B = 64); grown on demand.
_workspaces = {}


def _workspace(device: torch.device, stream: int, nbytes: int) -> Tensor:
    key = (device.index, stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _workspaces[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def _compute_loss_fwd(target: Tensor, pred: Tensor, antialias: bool = False) -> Tensor:
    """target (B,1,Ht,Wt) float32 or float64, pred (B,1,H,W) float32 -> (B,4): loss, ssim_term,
    l1, smooth.  A float64 target is rounded to float32 inside the kernel as it is read."""
    if not (target.is_cuda and pred.is_cuda):
        raise _lib.DepthwarpError("compute_loss: expected ROCm device tensors "
                                  "(the HIP kernels have no CPU path)")
    if target.dtype not in (torch.float32, torch.float64) or pred.dtype != torch.float32:
        raise TypeError(f"compute_loss: expected a float32 / float64 target and a float32 prediction, "
                        f"got {target.dtype} and {pred.dtype}")
    flags = (_lib.DW_CL_FLAG_ANTIALIAS if antialias else 0) | \
            (_lib.DW_CL_FLAG_TARGET_F64 if target.dtype == torch.float64 else 0)
    B, _, H, W = pred.shape
    if target.shape[0] != B:
        raise RuntimeError("compute_loss: target and pred disagree on the batch size")
    target = target.contiguous()
    pred = pred.contiguous()
    lib = _lib.lib()
    nbytes = lib.dw_compute_loss_workspace_bytes(B, H, W)
    if nbytes == 0:
        raise _lib.DepthwarpError(f"compute_loss: unsupported shape B={B} H={H} W={W}")
    dev = pred.device
    out = torch.empty((B, 4), dtype=torch.float32, device=dev)
    other = torch.cuda.current_device() != dev.index
    if other:  # the library launches on the current device
        prev = torch.cuda.current_device()
        torch.cuda.set_device(dev)
    try:
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = _workspace(dev, stream, nbytes)
        rc = lib.dw_compute_loss_fwd(B, target.shape[2], target.shape[3], H, W,
                                     target.data_ptr(), pred.data_ptr(), out.data_ptr(),
                                     flags, ws.data_ptr(), ws.numel(), stream)
    finally:
        if other:
            torch.cuda.set_device(prev)
    _lib.check(rc, "dw_compute_loss_fwd")
    return out


@torch.library.custom_op("depthwarp::compute_loss_fwd", mutates_args=())
def compute_loss_fwd(target: Tensor, pred: Tensor, antialias: bool = False) -> Tensor:
    """The same as a torch operator (traceable: `register_fake` below)."""
    return _compute_loss_fwd(target, pred, antialias)


@compute_loss_fwd.register_fake
def _(target, pred, antialias=False):
    return torch.empty((pred.shape[0], 4), dtype=torch.float32, device=pred.device)


def _terms(target: Tensor, pred: Tensor, height, width, antialias: bool) -> Tensor:
    """(B,4) tensor of loss, ssim_term, l1, smooth per sample."""
    with torch.no_grad():
        pred = _as_b1hw("pred", pred)
        target = _as_b1hw("target", target)
        if height is not None and tuple(pred.shape[-2:]) != (height, width):
            # the reference's `pred.view(1, 1, H, W)` raises on a size mismatch
            raise RuntimeError(f"pred has shape {tuple(pred.shape)}, expected (*,1,{height},{width})")
        # eager calls skip the operator dispatcher; under torch.compile the registered operator
        # is what gets traced
        fwd = compute_loss_fwd if torch.compiler.is_compiling() else _compute_loss_fwd
        # the reference hands over a float64 target (NumPy pipeline): the kernel reads it as it is
        # and rounds to float32 on the way in; anything else is cast to float32 first
        if target.dtype not in (torch.float32, torch.float64):
            target = target.float()
        return fwd(target, pred if pred.dtype == torch.float32 else pred.float(), antialias)


def compute_loss_terms(target: Tensor, pred: Tensor, height: int = None, width: int = None,
                       antialias: bool = False):
    """Per-sample terms as a dict of (B,) tensors: loss, ssim_term, l1, smooth."""
    out = _terms(target, pred, height, width, antialias)
    return {"loss": out[:, 0], "ssim_term": out[:, 1], "l1": out[:, 2], "smooth": out[:, 3]}


def compute_loss(target: Tensor, pred: Tensor, height: int = None, width: int = None,
                 antialias: bool = False) -> Tensor:
    """Drop-in for `Train.compute_loss(target, pred)`; the output size is
    `pred`'s (the reference reads it from `Config.height/width`).

    `antialias` selects how the ground truth is shrunk (train.py:276 calls
    torchvision's `resize`, whose tensor default is the plain bilinear resize in
    some releases and the anti-aliased one in others; the reference pins no
    version): False = two taps per axis, True = area-weighted triangle filter
    (`F.interpolate(..., antialias=True)`)."""
    out = _terms(target, pred, height, width, antialias)
    return out[0, 0] if out.shape[0] == 1 else out[:, 0]


def _one_map(name: str, t: Tensor) -> Tensor:
    """(1,H,W) or (1,1,H,W) -> the tensor itself if it is one contiguous map, else a contiguous copy."""
    if t.dim() not in (3, 4) or t.numel() != t.shape[-1] * t.shape[-2]:
        raise RuntimeError(f"{name}: expected one map, (1,H,W) or (1,1,H,W), got {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def compute_loss_lr(l_target: Tensor, l_pred: Tensor, r_target: Tensor, r_pred: Tensor,
                    height: int = None, width: int = None, antialias: bool = False,
                    return_terms: bool = False):
    """The reference's loss of a training step,
    `compute_loss(l_depth, l_pred) + compute_loss(r_depth, r_pred)` (train.py:113, 177), as one
    call: both views go through one launch pair (`dw_compute_loss_pair_fwd`) straight from the four
    tensors, nothing is stacked or copied, and each view's terms are bit-identical to a
    `compute_loss` call of its own.  Returns the 0-dim sum; with `return_terms` also the (2, 4)
    tensor of loss, ssim_term, l1, smooth per view."""
    with torch.no_grad():
        lt, lp = _one_map("l_target", l_target), _one_map("l_pred", l_pred)
        rt, rp = _one_map("r_target", r_target), _one_map("r_pred", r_pred)
        if not (lt.is_cuda and lp.is_cuda and rt.is_cuda and rp.is_cuda):
            raise _lib.DepthwarpError("compute_loss_lr: expected ROCm device tensors "
                                      "(the HIP kernels have no CPU path)")
        H, W = lp.shape[-2], lp.shape[-1]
        if tuple(rp.shape[-2:]) != (H, W) or (height is not None and (H, W) != (height, width)):
            # the reference's `pred.view(1, 1, H, W)` raises on a size mismatch
            raise RuntimeError(f"predictions have shapes {tuple(lp.shape)} and {tuple(rp.shape)}, expected (1,{height or H},{width or W})")
        if tuple(rt.shape[-2:]) != tuple(lt.shape[-2:]):
            raise RuntimeError("compute_loss_lr: the two ground-truth maps differ in size")
        if lp.dtype != torch.float32:
            lp = lp.float()
        if rp.dtype != torch.float32:
            rp = rp.float()
        if lt.dtype not in (torch.float32, torch.float64):
            lt = lt.float()
        if rt.dtype != lt.dtype:
            rt = rt.to(lt.dtype)
        flags = (_lib.DW_CL_FLAG_ANTIALIAS if antialias else 0) | \
                (_lib.DW_CL_FLAG_TARGET_F64 if lt.dtype == torch.float64 else 0)
        lib = _lib.lib()
        nbytes = lib.dw_compute_loss_workspace_bytes(2, H, W)
        if nbytes == 0:
            raise _lib.DepthwarpError(f"compute_loss_lr: unsupported shape H={H} W={W}")
        dev = lp.device
        out = torch.empty(9, dtype=torch.float32, device=dev)
        other = torch.cuda.current_device() != dev.index
        if other:  # the library launches on the current device
            prev = torch.cuda.current_device()
            torch.cuda.set_device(dev)
        try:
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = _workspace(dev, stream, nbytes)
            rc = lib.dw_compute_loss_pair_fwd(lt.shape[-2], lt.shape[-1], H, W, lt.data_ptr(), lp.data_ptr(),
                                              rt.data_ptr(), rp.data_ptr(), out.data_ptr(), flags,
                                              ws.data_ptr(), ws.numel(), stream)
        finally:
            if other:
                torch.cuda.set_device(prev)
        _lib.check(rc, "dw_compute_loss_pair_fwd")
        return (out[8], out[:8].view(2, 4)) if return_terms else out[8]


def compute_loss_into(out: Tensor, target: Tensor, pred: Tensor, antialias: bool = False) -> Tensor:
    """`compute_loss_terms` for callers that must not allocate (a loop under HIP graph capture,
    a harness that forbids device allocation): the caller passes the result buffer `out` (B, 4)
    float32 -- loss, ssim_term, l1, smooth per sample -- and the library uses its own fixed scratch
    array for the partial sums (`dw_compute_loss_fwd_static`).  Inputs must already be float32,
    contiguous (B,1,Ht,Wt) / (B,1,H,W) device tensors: nothing is converted or copied here.
    Problems beyond the scratch array's size raise; calls that overlap on one GPU race."""
    for name, t in (("out", out), ("target", target), ("pred", pred)):
        if not t.is_cuda:
            raise _lib.DepthwarpError(f"compute_loss_into: {name} must be a ROCm device tensor")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError(f"compute_loss_into: {name} must be contiguous float32")
    if pred.dim() != 4 or target.dim() != 4 or pred.shape[1] != 1 or target.shape[1] != 1:
        raise RuntimeError("compute_loss_into: expected target (B,1,Ht,Wt) and pred (B,1,H,W)")
    B, _, H, W = pred.shape
    if target.shape[0] != B or tuple(out.shape) != (B, 4):
        raise RuntimeError("compute_loss_into: target, pred and out (B,4) disagree on the batch size")
    with torch.cuda.device(pred.device):
        rc = _lib.lib().dw_compute_loss_fwd_static(B, target.shape[2], target.shape[3], H, W, target.data_ptr(),
                                                   pred.data_ptr(), out.data_ptr(),
                                                   _lib.DW_CL_FLAG_ANTIALIAS if antialias else 0,
                                                   torch.cuda.current_stream(pred.device).cuda_stream)
    _lib.check(rc, "dw_compute_loss_fwd_static")
    return out
