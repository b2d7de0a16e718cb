"""torch custom ops over libdepthwarp.so.

`depthwarp::photometric_fwd` / `depthwarp::photometric_bwd` wrap the fused
view-synthesis + photometric-loss kernels; `photometric_stack(...)` is the
differentiable user-facing call.  The ops run on ROCm devices only and raise
if the library is missing: there is no eager or CPU fallback.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import DwDesc, DW_NUM_SRC, DW_FLAG_COLOR_GRAD

__all__ = ["photometric_stack", "photometric_forward", "workspace_bytes", "PhotometricPlan", "PlanLoss"]


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check_f32(name: str, t: Tensor, shape=None):
    if not t.is_cuda:
        raise _lib.DepthwarpError(f"{name}: expected a ROCm device tensor, got {t.device} "
                                  "(the HIP kernels have no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous NCHW tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def _desc_from(disp: Sequence[Tensor], color: Sequence[Tensor]) -> DwDesc:
    B, C3, H, W = color[0].shape
    return DwDesc(B, H, W, len(disp), DW_NUM_SRC, 0)


def workspace_bytes(B: int, H: int, W: int, S: int = 4, flags: int = 0) -> int:
    d = DwDesc(B, H, W, S, DW_NUM_SRC, flags)
    n = _lib.lib().dw_photometric_workspace_bytes(d)
    if n == 0:
        raise _lib.DepthwarpError(f"unsupported photometric-stack shape B={B} H={H} W={W} S={S}")
    return n


def _validate(disp, color, color_pyr, K, inv_K, T, tie_noise):
    if len(color) != 3:
        raise ValueError("color must be [target, source(-1), source(+1)]")
    S = len(disp)
    if len(color_pyr) != S:
        raise ValueError("color_pyr needs one entry per scale")
    B, C3, H, W = color[0].shape
    if C3 != 3:
        raise ValueError("colour images must have 3 channels")
    for i, c in enumerate(color):
        _check_f32(f"color[{i}]", c, (B, 3, H, W))
    for s in range(S):
        _check_f32(f"disp[{s}]", disp[s], (B, 1, H >> s, W >> s))
        _check_f32(f"color_pyr[{s}]", color_pyr[s], (B, 3, H >> s, W >> s))
    _check_f32("K", K, (B, 4, 4))
    _check_f32("inv_K", inv_K, (B, 4, 4))
    _check_f32("T", T, (B, DW_NUM_SRC, 4, 4))
    if tie_noise is not None:
        _check_f32("tie_noise", tie_noise, (B, DW_NUM_SRC, H, W))


@torch.library.custom_op("depthwarp::photometric_fwd", mutates_args=())
def photometric_fwd(disp: Sequence[Tensor], color: Sequence[Tensor], color_pyr: Sequence[Tensor],
                    K: Tensor, inv_K: Tensor, T: Tensor, tie_noise: Optional[Tensor],
                    with_grad: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """Returns (loss_terms (1+2S,), mask (S,B,H,W) uint8, workspace bytes)."""
    _validate(disp, color, color_pyr, K, inv_K, T, tie_noise)
    desc = _desc_from(disp, color)
    dev = color[0].device
    S = desc.S
    nbytes = workspace_bytes(desc.B, desc.H, desc.W, S)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    loss = torch.empty(1 + 2 * S, dtype=torch.float32, device=dev)
    mask = torch.empty((S, desc.B, desc.H, desc.W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().dw_photometric_fwd(
            desc, _lib.ptr_array(disp), _lib.ptr_array(color), _lib.ptr_array(color_pyr),
            K.data_ptr(), inv_K.data_ptr(), T.data_ptr(),
            None if tie_noise is None else tie_noise.data_ptr(),
            _lib.ptr_array([mask[s] for s in range(S)]), loss.data_ptr(),
            ws.data_ptr(), nbytes, 1 if with_grad else 0, _stream(dev))
    _lib.check(rc, "dw_photometric_fwd")
    return loss, mask, ws


@photometric_fwd.register_fake
def _(disp, color, color_pyr, K, inv_K, T, tie_noise, with_grad):
    B, _, H, W = color[0].shape
    S = len(disp)
    dev = color[0].device
    return (torch.empty(1 + 2 * S, dtype=torch.float32, device=dev),
            torch.empty((S, B, H, W), dtype=torch.uint8, device=dev),
            torch.empty(workspace_bytes(B, H, W, S), dtype=torch.uint8, device=dev))


@torch.library.custom_op("depthwarp::photometric_bwd", mutates_args=())
def photometric_bwd(disp: Sequence[Tensor], K: Tensor, grad_loss: Tensor, workspace: Tensor,
                    height: int, width: int) -> Tuple[Tensor, Tensor]:
    """Returns (grad of every disparity map, concatenated flat; grad_T (B,F,4,4))."""
    S = len(disp)
    B = disp[0].shape[0]
    desc = DwDesc(B, height, width, S, DW_NUM_SRC, 0)
    dev = disp[0].device
    _check_f32("grad_loss", grad_loss)
    if grad_loss.numel() not in (1, 1 + 2 * S):
        raise ValueError(f"grad_loss: expected 1 or {1 + 2 * S} elements, got {grad_loss.numel()}")
    sizes = [d.numel() for d in disp]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    views, o = [], 0
    for n in sizes:
        views.append(flat[o:o + n])
        o += n
    grad_T = torch.empty((B, DW_NUM_SRC, 4, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().dw_photometric_bwd(
            desc, _lib.ptr_array(disp), K.data_ptr(), grad_loss.data_ptr(), grad_loss.numel(),
            workspace.data_ptr(), workspace.numel(), _lib.ptr_array(views),
            grad_T.data_ptr(), _stream(dev))
    _lib.check(rc, "dw_photometric_bwd")
    return flat, grad_T


@photometric_bwd.register_fake
def _(disp, K, grad_loss, workspace, height, width):
    dev = disp[0].device
    return (torch.empty(sum(d.numel() for d in disp), dtype=torch.float32, device=dev),
            torch.empty((disp[0].shape[0], DW_NUM_SRC, 4, 4), dtype=torch.float32, device=dev))


def _setup_context(ctx, inputs, output):
    disp, color, color_pyr, K, inv_K, T, tie_noise, with_grad = inputs
    _, _, ws = output
    ctx.with_grad = with_grad
    ctx.S = len(disp)
    ctx.hw = tuple(color[0].shape[-2:])
    ctx.n_color = len(color)
    ctx.n_pyr = len(color_pyr)
    ctx.save_for_backward(*disp, K, ws)


def _backward(ctx, g_loss, g_mask, g_ws):
    if not ctx.with_grad:
        raise RuntimeError("photometric_fwd was called with with_grad=False; "
                           "no backward state was kept")
    saved = ctx.saved_tensors
    disp, K, ws = list(saved[:ctx.S]), saved[ctx.S], saved[ctx.S + 1]
    # g_loss has 1 + 2S entries: the upstream gradient of the total and of every per-scale term
    flat, grad_T = photometric_bwd(disp, K, g_loss.contiguous(), ws, ctx.hw[0], ctx.hw[1])
    grads, o = [], 0
    for d in disp:
        grads.append(flat[o:o + d.numel()].view_as(d))
        o += d.numel()
    return (grads, [None] * ctx.n_color, [None] * ctx.n_pyr, None, None, grad_T, None, None)


photometric_fwd.register_autograd(_backward, setup_context=_setup_context)


def photometric_forward(disp, color, color_pyr, K, inv_K, T, tie_noise=None):
    """Returns (loss_terms, mask). loss_terms[0] is the total, [1:1+S] are L_photo
    per scale, [1+S:] L_smooth per scale (all differentiable). mask[s] packs the
    arg-min index (bits 0-1) and the validity of each source (bits 2, 3)."""
    need = torch.is_grad_enabled() and (any(d.requires_grad for d in disp) or T.requires_grad)
    terms, mask, _ = photometric_fwd(list(disp), list(color), list(color_pyr), K, inv_K, T,
                                     tie_noise, need)
    return terms, mask


def photometric_stack(disp, color, color_pyr, K, inv_K, T, tie_noise=None) -> Tensor:
    """Scalar multi-scale view-synthesis loss (differentiable w.r.t. disp and T)."""
    return photometric_forward(disp, color, color_pyr, K, inv_K, T, tie_noise)[0][0]


class PhotometricPlan:
    """Pre-allocated buffers + raw C-ABI calls for one problem shape.

    Nothing is allocated per call, so a step (forward + backward) can be
    replayed from a HIP graph or issued back to back from a tight loop.  The
    autograd path above allocates its outputs per call instead; both run the
    same kernels.

    With `flags=DW_FLAG_COLOR_GRAD` the forward pass also accumulates the
    gradient with respect to the two source images as the warp samples them
    (the scatter adjoint of the bilinear warp; the identity errors and the
    target image are not differentiated) and `backward_color()` returns it.
    """

    def __init__(self, B: int, H: int, W: int, S: int = 4, device="cuda", flags: int = 0):
        self.desc = DwDesc(B, H, W, S, DW_NUM_SRC, flags)
        self.device = torch.device(device)
        self.nbytes = workspace_bytes(B, H, W, S, flags)
        dev = self.device
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(1 + 2 * S, dtype=torch.float32, device=dev)
        self.mask = torch.empty((S, B, H, W), dtype=torch.uint8, device=dev)
        self.grad_disp = [torch.empty((B, 1, H >> s, W >> s), dtype=torch.float32, device=dev)
                          for s in range(S)]
        self.grad_T = torch.empty((B, DW_NUM_SRC, 4, 4), dtype=torch.float32, device=dev)
        self.one = torch.ones(1, dtype=torch.float32, device=dev)
        self._mask_ptrs = _lib.ptr_array([self.mask[s] for s in range(S)])
        self._gd_ptrs = _lib.ptr_array(self.grad_disp)
        self.generation = 0  # forward passes so far (lets an autograd node notice a later forward)
        self.grad_color = None
        if flags & DW_FLAG_COLOR_GRAD:
            self.grad_color = [torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
                               for _ in range(DW_NUM_SRC)]
            self._gc_ptrs = _lib.ptr_array(self.grad_color)
        self._lib = _lib.lib()

    def bind(self, disp, color, color_pyr, K, inv_K, T, tie_noise=None):
        """Validate once and keep the pointer arrays of a fixed set of inputs."""
        _validate(disp, color, color_pyr, K, inv_K, T, tie_noise)
        self._in = (list(disp), list(color), list(color_pyr), K, inv_K, T, tie_noise)
        self._p = (_lib.ptr_array(disp), _lib.ptr_array(color), _lib.ptr_array(color_pyr),
                   K.data_ptr(), inv_K.data_ptr(), T.data_ptr(),
                   None if tie_noise is None else tie_noise.data_ptr())
        return self

    def tile_kernel_only(self, enable: bool = True):
        """Profiling: later forward calls launch only the dominant (per-tile) kernel.  Call after
        one complete forward pass on the bound inputs: the tile kernel reads the depth maps that
        pass has left in the workspace."""
        if enable:
            self.desc.flags |= _lib.DW_FLAG_TILE_KERNEL_ONLY
        else:
            self.desc.flags &= ~_lib.DW_FLAG_TILE_KERNEL_ONLY
        return self

    def forward(self, with_grad: bool = True, stream: Optional[int] = None):
        d, c, py, K, iK, T, tie = self._p
        # the host side sizes its launch from the current device, and the stream belongs to it
        with torch.cuda.device(self.device):
            rc = self._lib.dw_photometric_fwd(self.desc, d, c, py, K, iK, T, tie, self._mask_ptrs,
                                              self.loss.data_ptr(), self.ws.data_ptr(), self.nbytes,
                                              1 if with_grad else 0,
                                              _stream(self.device) if stream is None else stream)
        _lib.check(rc, "dw_photometric_fwd")
        return self.loss

    def backward(self, grad_loss: Optional[Tensor] = None, stream: Optional[int] = None):
        """`grad_loss`: one element (upstream gradient of the total) or 1 + 2S (of every loss term)."""
        g = self.one if grad_loss is None else grad_loss
        with torch.cuda.device(self.device):
            rc = self._lib.dw_photometric_bwd(self.desc, self._p[0], self._p[3], g.data_ptr(), g.numel(),
                                              self.ws.data_ptr(), self.nbytes, self._gd_ptrs,
                                              self.grad_T.data_ptr(),
                                              _stream(self.device) if stream is None else stream)
        _lib.check(rc, "dw_photometric_bwd")
        return self.grad_disp, self.grad_T

    def backward_color(self, grad_loss: Optional[Tensor] = None, stream: Optional[int] = None):
        """d loss / d (color[1], color[2]) through the warp; needs flags=DW_FLAG_COLOR_GRAD
        and a preceding forward(with_grad=True)."""
        if self.grad_color is None:
            raise _lib.DepthwarpError("this plan was built without DW_FLAG_COLOR_GRAD")
        g = self.one if grad_loss is None else grad_loss
        with torch.cuda.device(self.device):
            rc = self._lib.dw_photometric_bwd_color(self.desc, g.data_ptr(), self.ws.data_ptr(), self.nbytes,
                                                    self._gc_ptrs,
                                                    _stream(self.device) if stream is None else stream)
        _lib.check(rc, "dw_photometric_bwd_color")
        return self.grad_color


class _PlanFunction(torch.autograd.Function):
    """autograd node over a bound PhotometricPlan: forward and backward are the raw C-ABI
    calls on the plan's pre-allocated buffers."""

    @staticmethod
    def forward(ctx, plan, need_grad, T, *disp):
        plan.forward(with_grad=need_grad)
        plan.generation += 1
        ctx.plan, ctx.generation, ctx.need_grad = plan, plan.generation, need_grad
        ctx.mark_non_differentiable(plan.mask)
        return plan.loss.clone(), plan.mask

    @staticmethod
    def backward(ctx, g_loss, _g_mask):
        plan = ctx.plan
        if not ctx.need_grad:
            raise RuntimeError("the loss was evaluated without gradients; no backward state was kept")
        if plan.generation != ctx.generation:
            raise RuntimeError("the plan ran another forward pass since this loss was computed: its "
                               "workspace holds the later call (one loss per plan and step)")
        grad_disp, grad_T = plan.backward(g_loss.contiguous())
        return (None, None, grad_T, *grad_disp)


class PlanLoss:
    """`photometric_stack` with pre-allocated state: the same call signature, but the
    workspace, masks and gradient buffers belong to a PhotometricPlan per problem shape and
    are reused by every step (nothing of that size is allocated per call, and a step is the
    two raw library calls).  The returned gradients alias the plan's buffers: consume them --
    as autograd does when it runs the networks' backward -- before the next forward pass of the
    same shape.  One loss per shape may be pending at a time."""

    def __init__(self):
        self._plans = {}
        self.last_mask = None

    def terms(self, disp, color, color_pyr, K, inv_K, T, tie_noise=None):
        B, _, H, W = color[0].shape
        key = (B, H, W, len(disp), color[0].device)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = PhotometricPlan(B, H, W, len(disp), color[0].device)
        plan.bind([d.detach() for d in disp], color, color_pyr, K, inv_K, T.detach(), tie_noise)
        need = torch.is_grad_enabled() and (any(d.requires_grad for d in disp) or T.requires_grad)
        terms, mask = _PlanFunction.apply(plan, need, T, *disp)
        self.last_mask = mask
        return terms

    def __call__(self, disp, color, color_pyr, K, inv_K, T, tie_noise=None) -> Tensor:
        return self.terms(disp, color, color_pyr, K, inv_K, T, tie_noise)[0]
