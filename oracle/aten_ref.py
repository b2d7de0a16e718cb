"""The same stack stated with stock ATen operators (matmul, F.interpolate,
F.grid_sample, ReflectionPad2d + avg_pool2d, torch.min), the way the public
formulation (Godard et al., ICCV 2019) is usually written.

TEST INFRASTRUCTURE ONLY.  Used to show that `oracle.photometric` (the
exact-order restatement the kernels are held to) is the same function: the two
agree to rounding in float32 and to ~1e-12 in float64, values and gradients.
It is not used for bit-exact comparisons, because these operators do not fix
their association order.  **Parity unpinned** (the reference has no such code).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def disp_to_depth(disp, min_depth=0.1, max_depth=100.0):
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


class BackprojectDepth(nn.Module):
    def __init__(self, batch_size, height, width, dtype=torch.float32):
        super().__init__()
        self.B, self.H, self.W = batch_size, height, width
        ys, xs = torch.meshgrid(torch.arange(height, dtype=dtype),
                                torch.arange(width, dtype=dtype), indexing="ij")
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], 0)[None].repeat(batch_size, 1, 1)
        ones = torch.ones(batch_size, 1, height * width, dtype=dtype)
        self.register_buffer("ones", ones)
        self.register_buffer("pix_coords", torch.cat([pix, ones], 1))

    def forward(self, depth, inv_K):
        cam = torch.matmul(inv_K[:, :3, :3], self.pix_coords)
        cam = depth.view(self.B, 1, -1) * cam
        return torch.cat([cam, self.ones], 1)


class Project3D(nn.Module):
    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.B, self.H, self.W, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        P = torch.matmul(K, T)[:, :3, :]
        cam = torch.matmul(P, points)
        pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + self.eps)
        pix = pix.view(self.B, 2, self.H, self.W).permute(0, 2, 3, 1).clone()
        pix[..., 0] /= self.W - 1
        pix[..., 1] /= self.H - 1
        return (pix - 0.5) * 2, cam[:, 2, :].view(self.B, self.H, self.W)


class SSIM(nn.Module):
    def __init__(self):
        super().__init__()
        self.pool = nn.AvgPool2d(3, 1)
        self.refl = nn.ReflectionPad2d(1)
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, x, y):
        x = self.refl(x)
        y = self.refl(y)
        mu_x = self.pool(x)
        mu_y = self.pool(y)
        sigma_x = self.pool(x ** 2) - mu_x ** 2
        sigma_y = self.pool(y ** 2) - mu_y ** 2
        sigma_xy = self.pool(x * y) - mu_x * mu_y
        n = (2 * mu_x * mu_y + self.C1) * (2 * sigma_xy + self.C2)
        d = (mu_x ** 2 + mu_y ** 2 + self.C1) * (sigma_x + sigma_y + self.C2)
        return torch.clamp((1 - n / d) / 2, 0, 1)


def get_smooth_loss(disp, img):
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    gdx = gdx * torch.exp(-gix)
    gdy = gdy * torch.exp(-giy)
    return gdx.mean() + gdy.mean()


def photometric_stack(disp, color, color_pyr, K, inv_K, T, tie_noise=None,
                      return_aux=False):
    S = len(disp)
    target = color[0]
    B, _, H, W = target.shape
    nF = T.shape[1]
    dt = target.dtype
    backproject = BackprojectDepth(B, H, W, dtype=dt)
    project = Project3D(B, H, W)
    ssim = SSIM()

    def reproj_err(pred, tgt):
        l1 = (tgt - pred).abs().mean(1, True)
        return 0.85 * ssim(pred, tgt).mean(1, True) + 0.15 * l1

    ident = []
    for f in range(nF):
        e = reproj_err(color[1 + f], target)
        if tie_noise is not None:
            e = e + tie_noise[:, f:f + 1]
        ident.append(e)

    total = 0.0
    aux = {"L_photo": [], "L_sm": [], "idx": [], "valid": []}
    for s in range(S):
        d = disp[s]
        if d.shape[-2:] != (H, W):
            d = F.interpolate(d, (H, W), mode="bilinear", align_corners=False)
        _, depth = disp_to_depth(d)
        cam = backproject(depth, inv_K)
        errs, valids = [], []
        for f in range(nF):
            grid, z = project(cam, K, T[:, f])
            warped = F.grid_sample(color[1 + f], grid, mode="bilinear",
                                   padding_mode="border", align_corners=True)
            errs.append(reproj_err(warped, target))
            valids.append((grid[..., 0] >= -1) & (grid[..., 0] <= 1)
                          & (grid[..., 1] >= -1) & (grid[..., 1] <= 1) & (z > 0))
        combined = torch.cat(ident + errs, 1)
        m, idx = torch.min(combined, 1)
        L_photo = m.mean()
        mean_disp = disp[s].mean(2, True).mean(3, True)
        L_sm = get_smooth_loss(disp[s] / (mean_disp + 1e-7), color_pyr[s])
        total = total + L_photo + 1e-3 * L_sm / (2 ** s)
        aux["L_photo"].append(L_photo.detach())
        aux["L_sm"].append(L_sm.detach())
        aux["idx"].append(idx)
        aux["valid"].append(torch.stack(valids, 1))
    loss = total / S
    return (loss, aux) if return_aux else loss
