#!/usr/bin/env python3
"""Writes the golden fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference (/root/reference) holds no tests, fixtures or golden vectors and
cannot be imported here (torchvision, kornia, cv2, skimage, tensorboardX are
absent; its model constructor fetches weights over the network), so these
vectors pin the build-owned oracle against drift; they do not pin it to the
reference ("parity unpinned").  Inputs are regenerated from the seeds with
`depth_estimation_amd.synth.make_sequence`; only expected outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from depth_estimation_amd.synth import make_sequence  # noqa: E402
from oracle import photometric as ph  # noqa: E402
from oracle import compute_loss as cl  # noqa: E402

PHOTO_CASES = [dict(B=2, H=24, W=80, seed=0, tie=False), dict(B=2, H=48, W=160, seed=0, tie=False),
               dict(B=1, H=32, W=96, seed=5, tie=True)]
CL_CASES = [dict(B=1, Ht=375, Wt=1242, H=64, W=128, seed=0), dict(B=1, Ht=375, Wt=1242, H=256, W=512, seed=1),
            dict(B=2, Ht=64, Wt=128, H=64, W=128, seed=2), dict(B=1, Ht=100, Wt=300, H=48, W=160, seed=3),
            # the anti-aliased resize of the ground truth (the other torchvision default)
            dict(B=1, Ht=375, Wt=1242, H=256, W=512, seed=1, aa=True),
            dict(B=2, Ht=100, Wt=300, H=48, W=160, seed=4, aa=True)]


def photo_case(c):
    seq = make_sequence(c["B"], c["H"], c["W"], seed=c["seed"], tie_noise=c["tie"])
    disp = [d.clone().requires_grad_(True) for d in seq.disp]
    T = seq.T.clone().requires_grad_(True)
    loss, aux = ph.photometric_stack(disp, seq.color, seq.color_pyr, seq.K, seq.inv_K, T,
                                     seq.tie_noise, return_aux=True)
    loss.backward()
    out = {"loss": loss.detach().numpy(), "grad_T": T.grad.numpy(),
           "checksum_inputs": np.float64(sum(float(t.double().sum()) for t in seq.disp + seq.color))}
    for s in range(len(disp)):
        out[f"L_photo{s}"] = aux["L_photo"][s].numpy()
        out[f"L_sm{s}"] = aux["L_sm"][s].numpy()
        out[f"mask{s}"] = ph.pack_mask(aux["idx"][s], aux["valid"][s]).numpy()
        out[f"grad_disp{s}"] = disp[s].grad.numpy()
        out[f"min_gap{s}"] = np.float32(aux["gap"][s].min())
    return out


COLOR_CASES = [dict(B=1, H=32, W=96, seed=9)]


def color_case(c):
    """Gradient of the loss w.r.t. the two source images as the warp samples them."""
    seq = make_sequence(c["B"], c["H"], c["W"], seed=c["seed"])
    warp_src = [seq.color[1 + f].clone().requires_grad_(True) for f in range(2)]
    loss = ph.photometric_stack(seq.disp, seq.color, seq.color_pyr, seq.K, seq.inv_K, seq.T,
                                warp_src=warp_src)
    loss.backward()
    return {"loss": loss.detach().numpy(), "grad_color0": warp_src[0].grad.numpy(),
            "grad_color1": warp_src[1].grad.numpy()}


def cl_inputs(c):
    g = torch.Generator().manual_seed(c["seed"])
    target = torch.rand(c["B"], c["Ht"], c["Wt"], generator=g)
    pred = torch.rand(c["B"], c["H"], c["W"], generator=g)
    return target, pred


def write_compute_loss():
    rows = {}
    for i, c in enumerate(CL_CASES):
        t, p = cl_inputs(c)
        terms = cl.compute_loss_terms(t, p, antialias=c.get("aa", False))
        rows[f"case{i}"] = np.stack([terms[k].numpy() for k in ("loss", "ssim_term", "l1", "smooth")], 1)
    np.savez_compressed(os.path.join(HERE, "compute_loss.npz"), **rows)
    print("wrote compute_loss.npz")


# ---- the stages next to the headline path: soft-argmin, cost volume, metrics, LiDAR projection
def next_inputs():
    """Seeded inputs of the `next_rows.npz` fixture (regenerated, not stored)."""
    from depth_estimation_amd.lidar import KITTI_LIKE_CALIB, synthetic_scan, velo_to_image_matrix
    g = torch.Generator().manual_seed(21)
    cost = torch.randn(1, 16, 6, 10, generator=g) * 3.0
    maps = [torch.randn(1, 2, 4, 12, generator=g) for _ in range(4)]
    gt = torch.rand(2, 1, 20, 36, generator=g) * 80.0
    pred = torch.rand(2, 12, 20, generator=g) * 0.5 + 1e-4
    img = torch.rand(2, 3, 9, 11, generator=g) * 5.0 - 1.0
    shape = (47, 156)
    P = np.diag([shape[1] / 1242.0, shape[0] / 375.0, 1.0]) @ velo_to_image_matrix(
        KITTI_LIKE_CALIB["cam2cam"], KITTI_LIKE_CALIB["velo2cam"], 2)
    scan = synthetic_scan(2500, seed=7)
    return {"cost": cost, "maps": maps, "gt": gt, "pred": pred, "img": img, "scan": scan, "P": P, "shape": shape}


def next_outputs(inp):
    from oracle.cost_volume import cost_volume
    from oracle.lidar import project_scan
    from oracle.metrics import compute_errors_one_side, normalize_image
    from oracle.soft_argmin import soft_argmin
    cost = inp["cost"].clone().requires_grad_(True)
    disp = soft_argmin(cost)
    w = torch.linspace(-1.0, 2.0, disp.numel()).view_as(disp)
    (disp * w).sum().backward()
    left, right = cost_volume(*inp["maps"], maxdisp=5)
    pts, depth = project_scan(inp["scan"], inp["P"], inp["shape"])
    return {"soft_argmin": disp.detach().numpy(), "soft_argmin_grad": cost.grad.numpy(),
            "cost_left": left.numpy(), "cost_right": right.numpy(),
            "rmse": compute_errors_one_side(inp["gt"], inp["pred"], 12, 20).numpy(),
            "normalized": normalize_image(inp["img"]).numpy(),
            "lidar_pts": pts, "lidar_depth": depth}


def write_next_rows():
    np.savez_compressed(os.path.join(HERE, "next_rows.npz"), **next_outputs(next_inputs()))
    print("wrote next_rows.npz")


def main():
    write_next_rows()
    for c in PHOTO_CASES:
        name = f"photo_B{c['B']}_{c['H']}x{c['W']}_s{c['seed']}{'_tie' if c['tie'] else ''}.npz"
        np.savez_compressed(os.path.join(HERE, name), **photo_case(c))
        print("wrote", name)
    for c in COLOR_CASES:
        name = f"colorgrad_B{c['B']}_{c['H']}x{c['W']}_s{c['seed']}.npz"
        np.savez_compressed(os.path.join(HERE, name), **color_case(c))
        print("wrote", name)
    write_compute_loss()


if __name__ == "__main__":
    main()
