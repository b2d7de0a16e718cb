"""Exact-order CPU oracle of the view-synthesis + photometric-loss stack.

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).  **Parity unpinned**: the
reference (`/root/reference`) contains none of these stages; the only datum it
offers is the normalised intrinsics matrix at `dataset/load_dataset.py:21-24`.

Every stage is written with element-wise torch operators only (add, sub, mul,
div, floor, clamp, compare, index), one IEEE-754 rounding per operator, in a
FIXED association order.  No matmul, no `grid_sample`, no pooling, no fused
multiply-add: those kernels are free to re-associate or contract, which would
make validity masks and arg-min indices irreproducible bit for bit.  The HIP
kernels evaluate the same expressions in the same order with contraction off,
so masks, indices and the per-pixel error maps agree exactly; the scalar loss
and the gradients (autograd through this file) agree to rounding.

Stage numbering (S1..S10) in the comments:

  S1  disparity bilinear up-sample to full resolution (align_corners=False)
  S2  disparity -> depth
  S3  back-projection with inv_K
  S4  projection with P = (K @ T)[:3]  (+ validity mask S4v)
  S5  bilinear warp, border padding, align_corners=True
  S6  photometric error 0.85*SSIM(3x3, reflect) + 0.15*L1
  S7  identity (un-warped) photometric error (+ optional tie noise)
  S8  per-pixel minimum over [id(-1), id(+1), reproj(-1), reproj(+1)], auto-mask
  S9  edge-aware smoothness on the mean-normalised disparity
  S10 L = 1/S * sum_s (L_photo[s] + 1e-3 * L_sm[s] / 2^s)

Means over small fixed counts are evaluated as  sum * fl(1/n)  (n = 9 for the
3x3 window, n = 3 for the colour channels); the 3x3 window sum is vertical
first ((r0 + r1) + r2 per column) and then horizontal ((c0 + c1) + c2).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

MIN_DEPTH = 0.1
MAX_DEPTH = 100.0
SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2
W_SSIM = 0.85
W_L1 = 0.15
W_SMOOTH = 1e-3
PROJ_EPS = 1e-7
MEAN_EPS = 1e-7


def _c(v, ref):
    """Python scalar -> 0-dim tensor of ref's dtype (one rounding, double -> dtype)."""
    return torch.tensor(v, dtype=ref.dtype, device=ref.device)


# --------------------------------------------------------------------------- S1
def _axis_taps(out_size: int, in_size: int, dtype, device):
    scale = _c(in_size, torch.empty((), dtype=dtype, device=device)) / out_size
    dst = torch.arange(out_size, dtype=dtype, device=device)
    src = (dst + 0.5) * scale - 0.5
    src = torch.clamp(src, min=0.0)
    i0 = src.floor().long()
    i1 = i0 + (i0 < in_size - 1).long()
    l1 = src - i0.to(dtype)
    l0 = 1.0 - l1
    return i0, i1, l0, l1


def upsample_bilinear(d: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """(B,1,h,w) -> (B,1,H,W); same formula as `F.interpolate(mode='bilinear',
    align_corners=False)`:  l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11)."""
    h, w = d.shape[-2:]
    if (h, w) == (H, W):
        return d
    y0, y1, ly0, ly1 = _axis_taps(H, h, d.dtype, d.device)
    x0, x1, lx0, lx1 = _axis_taps(W, w, d.dtype, d.device)
    v00 = d[:, :, y0[:, None], x0[None, :]]
    v01 = d[:, :, y0[:, None], x1[None, :]]
    v10 = d[:, :, y1[:, None], x0[None, :]]
    v11 = d[:, :, y1[:, None], x1[None, :]]
    top = lx0 * v00 + lx1 * v01
    bot = lx0 * v10 + lx1 * v11
    return ly0[:, None] * top + ly1[:, None] * bot


# --------------------------------------------------------------------------- S2
def disp_to_depth(disp, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH):
    min_disp = 1.0 / max_depth
    max_disp = 1.0 / min_depth
    scaled = _c(min_disp, disp) + _c(max_disp - min_disp, disp) * disp
    return scaled, 1.0 / scaled


# ------------------------------------------------------------------------ S3/S4
def pixel_rays(inv_K, H, W):
    """ray_i = (invK_i0*x + invK_i1*y) + invK_i2   -> three (B,H,W) tensors."""
    dt, dev = inv_K.dtype, inv_K.device
    x = torch.arange(W, dtype=dt, device=dev)[None, None, :]
    y = torch.arange(H, dtype=dt, device=dev)[None, :, None]
    rays = []
    for i in range(3):
        a0 = inv_K[:, i, 0][:, None, None]
        a1 = inv_K[:, i, 1][:, None, None]
        a2 = inv_K[:, i, 2][:, None, None]
        rays.append((a0 * x + a1 * y) + a2)
    return rays


def proj_matrix(K, T):
    """P_ij = ((K_i0*T_0j + K_i1*T_1j) + K_i2*T_2j) + K_i3*T_3j,  i < 3.  (B,3,4)"""
    rows = []
    for i in range(3):
        cols = []
        for j in range(4):
            v = ((K[:, i, 0] * T[:, 0, j] + K[:, i, 1] * T[:, 1, j])
                 + K[:, i, 2] * T[:, 2, j]) + K[:, i, 3] * T[:, 3, j]
            cols.append(v)
        rows.append(torch.stack(cols, -1))
    return torch.stack(rows, 1)


def project(depth, rays, P, H, W):
    """depth (B,H,W).  Returns u, v (pixel units), gx, gy (normalised), valid."""
    X, Y, Z = (depth * r for r in rays)
    c = []
    for i in range(3):
        p0, p1, p2, p3 = (P[:, i, j][:, None, None] for j in range(4))
        c.append(((p0 * X + p1 * Y) + p2 * Z) + p3)
    zp = c[2] + _c(PROJ_EPS, depth)
    u = c[0] / zp
    v = c[1] / zp
    gx = (u / (W - 1) - 0.5) * 2.0
    gy = (v / (H - 1) - 0.5) * 2.0
    valid = (gx >= -1) & (gx <= 1) & (gy >= -1) & (gy <= 1) & (c[2] > 0)
    return u, v, gx, gy, valid


# --------------------------------------------------------------------------- S5
def _clip_border(x, size):
    """Clamp to [0, size-1]; the gradient is 1 strictly inside and 0 on or
    outside the border (ATen `clip_coordinates_set_grad`)."""
    xc = torch.clamp(x, 0.0, float(size - 1))
    inside = (x > 0) & (x < size - 1)
    return torch.where(inside, x, xc.detach())


def warp_bilinear(src, gx, gy):
    """src (B,C,H,W); gx, gy (B,H,W) normalised.  Border padding,
    align_corners=True.  out = ((nw*w_nw + ne*w_ne) + sw*w_sw) + se*w_se."""
    B, C, H, W = src.shape
    ix = ((gx + 1.0) / 2.0) * (W - 1)
    iy = ((gy + 1.0) / 2.0) * (H - 1)
    ix = _clip_border(ix, W)
    iy = _clip_border(iy, H)
    ix0f = ix.detach().floor()
    iy0f = iy.detach().floor()
    tx = ix - ix0f
    ty = iy - iy0f
    ex = 1.0 - tx
    ey = 1.0 - ty
    x0 = ix0f.long()
    y0 = iy0f.long()
    x1 = torch.clamp(x0 + 1, max=W - 1)
    y1 = torch.clamp(y0 + 1, max=H - 1)
    flat = src.reshape(B, C, H * W)

    def tap(yy, xx):
        idx = (yy * W + xx).reshape(B, 1, H * W).expand(B, C, H * W)
        return torch.gather(flat, 2, idx).reshape(B, C, H, W)

    w_nw = (ex * ey)[:, None]
    w_ne = (tx * ey)[:, None]
    w_sw = (ex * ty)[:, None]
    w_se = (tx * ty)[:, None]
    return ((tap(y0, x0) * w_nw + tap(y0, x1) * w_ne) + tap(y1, x0) * w_sw) + tap(y1, x1) * w_se


# --------------------------------------------------------------------------- S6
def box3_reflect(a):
    """3x3 mean with 1-px reflection padding: vertical sums, then horizontal
    sums, then * fl(1/9)."""
    p = F.pad(a, (1, 1, 1, 1), mode="reflect")
    v = (p[:, :, 0:-2, :] + p[:, :, 1:-1, :]) + p[:, :, 2:, :]
    h = (v[..., 0:-2] + v[..., 1:-1]) + v[..., 2:]
    return h * _c(1.0 / 9.0, a)


def ssim_loss_map(x, y):
    """clamp((1 - SSIM(x, y)) / 2, 0, 1) per channel, (B,C,H,W)."""
    C1 = _c(SSIM_C1, x)
    C2 = _c(SSIM_C2, x)
    mu_x = box3_reflect(x)
    mu_y = box3_reflect(y)
    sig_x = box3_reflect(x * x) - mu_x * mu_x
    sig_y = box3_reflect(y * y) - mu_y * mu_y
    sig_xy = box3_reflect(x * y) - mu_x * mu_y
    n = (2.0 * mu_x * mu_y + C1) * (2.0 * sig_xy + C2)
    d = (mu_x * mu_x + mu_y * mu_y + C1) * (sig_x + sig_y + C2)
    return torch.clamp((1.0 - n / d) / 2.0, 0.0, 1.0)


def _mean3(a):
    """(B,3,H,W) -> (B,H,W):  ((a0 + a1) + a2) * fl(1/3)."""
    return ((a[:, 0] + a[:, 1]) + a[:, 2]) * _c(1.0 / 3.0, a)


def reprojection_error(pred, target):
    l1 = _mean3((pred - target).abs())
    ss = _mean3(ssim_loss_map(pred, target))
    return _c(W_SSIM, pred) * ss + _c(W_L1, pred) * l1


# --------------------------------------------------------------------------- S8
def min_first(cands):
    """Sequential strict-less-than scan: first minimal index wins ties."""
    m = cands[0]
    idx = torch.zeros_like(m, dtype=torch.uint8)
    second = torch.full_like(m, float("inf"))
    for k in range(1, len(cands)):
        lt = cands[k] < m
        second = torch.where(lt, m, torch.minimum(second, cands[k]))
        idx = torch.where(lt, torch.full_like(idx, k), idx)
        m = torch.where(lt, cands[k], m)
    return m, idx, (second - m).detach()


# --------------------------------------------------------------------------- S9
def smooth_loss(disp, img):
    """Edge-aware smoothness of the mean-normalised disparity. disp (B,1,h,w),
    img (B,3,h,w)."""
    mean = disp.mean(dim=(2, 3), keepdim=True)
    d = disp / (mean + MEAN_EPS)
    gdx = (d[..., :, :-1] - d[..., :, 1:]).abs()
    gdy = (d[..., :-1, :] - d[..., 1:, :]).abs()
    gix = (img[..., :, :-1] - img[..., :, 1:]).abs().mean(1, keepdim=True)
    giy = (img[..., :-1, :] - img[..., 1:, :]).abs().mean(1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


# --------------------------------------------------------------- whole stack
def photometric_stack(disp, color, color_pyr, K, inv_K, T, tie_noise=None,
                      return_aux=False, warp_src=None):
    """
    disp       list of S tensors (B,1,H/2^s,W/2^s)
    color      list [target, source(-1), source(+1)], each (B,3,H,W)
    color_pyr  list of S tensors, target colour at each scale ([0] is color[0])
    K, inv_K   (B,4,4) pixel-unit intrinsics
    T          (B,F,4,4) target->source transforms, F = 2
    tie_noise  optional (B,F,H,W) added to the identity errors
    warp_src   optional list of F tensors (B,3,H,W) sampled by the warp in place
               of color[1:]; with the same values as color[1:] the loss is the
               same, and its autograd gradient is the adjoint of the warp alone
               (the identity errors keep reading color[1:])
    Returns the scalar loss (and a dict of per-scale maps when return_aux).
    """
    S = len(disp)
    target = color[0]
    B, _, H, W = target.shape
    nF = T.shape[1]
    rays = pixel_rays(inv_K, H, W)
    P = [proj_matrix(K, T[:, f]) for f in range(nF)]

    ident = []
    for f in range(nF):
        e = reprojection_error(color[1 + f], target)
        if tie_noise is not None:
            e = e + tie_noise[:, f]
        ident.append(e)

    total = 0.0
    aux = {"L_photo": [], "L_sm": [], "idx": [], "valid": [], "gap": [], "E": []}
    for s in range(S):
        up = upsample_bilinear(disp[s], H, W)
        _, depth = disp_to_depth(up)
        depth = depth[:, 0]
        reproj, valids = [], []
        for f in range(nF):
            _, _, gx, gy, valid = project(depth, rays, P[f], H, W)
            warped = warp_bilinear(color[1 + f] if warp_src is None else warp_src[f], gx, gy)
            reproj.append(reprojection_error(warped, target))
            valids.append(valid)
        m, idx, gap = min_first(ident + reproj)
        L_photo = m.mean()
        L_sm = smooth_loss(disp[s], color_pyr[s])
        total = total + (L_photo + W_SMOOTH * L_sm / (2 ** s))
        if return_aux:
            aux["L_photo"].append(L_photo.detach())
            aux["L_sm"].append(L_sm.detach())
            aux["idx"].append(idx)
            aux["valid"].append(torch.stack(valids, 1))
            aux["gap"].append(gap)
            aux["E"].append(torch.stack([e.detach() for e in ident + reproj], 1))
    loss = total / S
    return (loss, aux) if return_aux else loss


def pack_mask(idx, valid):
    """Packed byte the kernels emit: bits 0-1 arg-min index, bit 2 / bit 3
    validity of source frame 0 / 1."""
    out = idx.to(torch.uint8)
    for f in range(valid.shape[1]):
        out = out | (valid[:, f].to(torch.uint8) << (2 + f))
    return out


# -------------------------------------------------- unfused module restatements
def backproject_depth(depth, inv_K):
    """depth (B,1,H,W) -> homogeneous camera points (B,4,H*W)."""
    B, _, H, W = depth.shape
    rays = pixel_rays(inv_K, H, W)
    pts = [(depth[:, 0] * r).reshape(B, 1, -1) for r in rays]
    pts.append(torch.ones_like(pts[0]))
    return torch.cat(pts, 1)


def project_3d(points, K, T, H, W, eps=PROJ_EPS):
    """points (B,4,N) -> sampling grid (B,H,W,2), (x, y) last."""
    B = points.shape[0]
    P = proj_matrix(K, T)
    X, Y, Z, Wh = (points[:, j] for j in range(4))
    c = []
    for i in range(3):
        p0, p1, p2, p3 = (P[:, i, j][:, None] for j in range(4))
        c.append(((p0 * X + p1 * Y) + p2 * Z) + p3 * Wh)
    zp = c[2] + _c(eps, points)
    gx = ((c[0] / zp) / (W - 1) - 0.5) * 2.0
    gy = ((c[1] / zp) / (H - 1) - 0.5) * 2.0
    return torch.stack([gx, gy], -1).reshape(B, H, W, 2)
