"""Velodyne scan -> image plane on the HIP path.

Mirrors `Dataset.generate_depth_map` of the reference
(/root/reference/dataset/load_dataset.py:99-153) and its helpers
`read_calib_file` (:167-186) and `load_velodyne_points` (:188-194).  The
projection, the in-bounds filter and the sparse nearest-depth map run in one
batched HIP call (`project_scans`); calibration parsing and the Delaunay hole
filling (`fill_holes`; the reference's `lin_interp`, :155-165, SciPy) stay on the host as in the reference.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib

__all__ = ["read_calib_file", "load_velodyne_points", "velo_to_image_matrix", "project_scans",
           "fill_holes", "generate_depth_map", "synthetic_scan", "KITTI_LIKE_CALIB"]


def read_calib_file(path: str) -> dict:
    """KITTI calibration text -> dict; numeric lines become float64 arrays (:167-186)."""
    data = {}
    with open(path, "r") as f:
        for line in f:
            if ":" not in line:
                continue
            key, value = line.split(":", 1)
            value = value.strip()
            try:
                data[key] = np.array([float(t) for t in value.split(" ")])
            except ValueError:
                data[key] = value
    return data


def load_velodyne_points(filename: str) -> np.ndarray:
    """KITTI scan file (little-endian float32 quadruples forward, left, up,
    reflectance) -> (N, 4) float32 rows with the reflectance replaced by the
    homogeneous 1, the array `generate_depth_map` of the reference works on (:188-194)."""
    raw = np.fromfile(filename, dtype="<f4")
    if raw.size % 4:
        raise ValueError(f"{filename}: {raw.size} floats is not a whole number of (x, y, z, r) records")
    scan = np.ascontiguousarray(raw.reshape(raw.size // 4, 4), dtype=np.float32)
    scan[:, 3] = 1.0
    return scan


def velo_to_image_matrix(cam2cam: dict, velo2cam: dict, cam: int = 2) -> np.ndarray:
    """P_velo2im (3, 4) float64 = P_rect_0<cam> . R_cam2rect . velo2cam (:104-114)."""
    v2c = np.eye(4)
    v2c[:3, :3] = np.asarray(velo2cam["R"], dtype=np.float64).reshape(3, 3)
    v2c[:3, 3] = np.asarray(velo2cam["T"], dtype=np.float64)
    rect = np.eye(4)
    rect[:3, :3] = np.asarray(cam2cam["R_rect_00"], dtype=np.float64).reshape(3, 3)
    P_rect = np.asarray(cam2cam["P_rect_0" + str(cam)], dtype=np.float64).reshape(3, 4)
    return np.dot(np.dot(P_rect, rect), v2c)


def project_scans(points: Tensor, n_points: Tensor, P: Tensor, im_shape: Tuple[int, int],
                  vel_depth: bool = False):
    """Batched :116-148.

    points (B, n_max, 4) float32 on the GPU, n_points (B) int32, P (B, 3, 4)
    float64, im_shape (H, W).  Returns (pts_im (B, n_max, 3) float64 with rows
    past n_valid zeroed, n_valid (B) int32, depth (B, H, W) float64).
    """
    if not points.is_cuda:
        raise _lib.DepthwarpError("project_scans: expected ROCm device tensors "
                                  "(the HIP kernels have no CPU path)")
    if points.dtype != torch.float32 or points.dim() != 3 or points.shape[2] != 4:
        raise ValueError(f"points: expected (B, n_max, 4) float32, got {tuple(points.shape)} {points.dtype}")
    B, n_max, _ = points.shape
    H, W = int(im_shape[0]), int(im_shape[1])
    dev = points.device
    points = points.contiguous()
    n_points = n_points.to(device=dev, dtype=torch.int32).contiguous()
    P = P.to(device=dev, dtype=torch.float64).contiguous()
    if tuple(P.shape) != (B, 3, 4) or tuple(n_points.shape) != (B,):
        raise ValueError("P must be (B, 3, 4) and n_points (B,)")
    lib = _lib.lib()
    nbytes = lib.dw_lidar_workspace_bytes(B, n_max, H, W)
    if nbytes == 0:
        raise _lib.DepthwarpError(f"project_scans: unsupported shape B={B} n_max={n_max} H={H} W={W}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pts_im = torch.zeros((B, n_max, 3), dtype=torch.float64, device=dev)
    n_valid = torch.empty(B, dtype=torch.int32, device=dev)
    depth = torch.empty((B, H, W), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dw_lidar_project(B, n_max, H, W, points.data_ptr(), n_points.data_ptr(), P.data_ptr(),
                                  int(bool(vel_depth)), pts_im.data_ptr(), n_valid.data_ptr(),
                                  depth.data_ptr(), ws.data_ptr(), nbytes,
                                  torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "dw_lidar_project")
    return pts_im, n_valid, depth


def fill_holes(shape, pts_im: np.ndarray) -> np.ndarray:
    """Dense map from the kept points (x, y, z) of one scan: piecewise-linear
    interpolation of z over the Delaunay triangulation of the (row, column)
    positions, 0 outside their convex hull -- the host-side stage that follows the
    projection in the reference (:151, :155-165).  Out of the GPU path: SciPy does it."""
    from scipy.interpolate import LinearNDInterpolator
    rows_cols = np.stack([pts_im[:, 1], pts_im[:, 0]], axis=1)
    surface = LinearNDInterpolator(rows_cols, pts_im[:, 2], fill_value=0.0)
    height, width = int(shape[0]), int(shape[1])
    queries = np.indices((height, width)).reshape(2, height * width).T
    return surface(queries).reshape(height, width)


def generate_depth_map(calib_path: str, velo_filename: str, interp: bool = True, cam: int = 2,
                       vel_depth: bool = False, device="cuda") -> np.ndarray:
    """Drop-in for `Dataset.generate_depth_map` (:99-153), one scan.

    The reference ends by replacing the sparse map with the Delaunay
    interpolation of the projected points regardless of `interp` (:151; its
    callers pass the camera id in that slot, :67-68); `interp=False` here
    returns the sparse map of :137-148 instead.
    """
    cam2cam = read_calib_file(os.path.join(calib_path, "calib_cam_to_cam.txt"))
    velo2cam = read_calib_file(os.path.join(calib_path, "calib_velo_to_cam.txt"))
    im_shape = cam2cam["S_rect_02"][::-1].astype(np.int32)
    P = velo_to_image_matrix(cam2cam, velo2cam, cam)
    velo = load_velodyne_points(velo_filename)
    pts = torch.from_numpy(velo)[None].to(device)
    pts_im, n_valid, depth = project_scans(pts, torch.tensor([len(velo)], dtype=torch.int32),
                                           torch.from_numpy(P)[None], (int(im_shape[0]), int(im_shape[1])),
                                           vel_depth)
    if not interp:
        return depth[0].cpu().numpy()
    xyd = pts_im[0, :int(n_valid[0])].cpu().numpy()
    return fill_holes((int(im_shape[0]), int(im_shape[1])), xyd)


# Calibration of the shape KITTI publishes (rectified camera 2, 1242 x 375).
KITTI_LIKE_CALIB = {
    "cam2cam": {
        "S_rect_02": np.array([1242.0, 375.0]),
        "R_rect_00": np.array([9.999239e-01, 9.837760e-03, -7.445048e-03, -9.869795e-03, 9.999421e-01,
                               -4.278459e-03, 7.402527e-03, 4.351614e-03, 9.999631e-01]),
        "P_rect_02": np.array([7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01, 0.0, 7.215377e+02,
                               1.728540e+02, 2.163791e-01, 0.0, 0.0, 1.0, 2.745884e-03]),
        "P_rect_03": np.array([7.215377e+02, 0.0, 6.095593e+02, -3.395242e+02, 0.0, 7.215377e+02,
                               1.728540e+02, 2.199936e+00, 0.0, 0.0, 1.0, 2.729905e-03]),
    },
    "velo2cam": {
        "R": np.array([7.533745e-03, -9.999714e-01, -6.166020e-04, 1.480249e-02, 7.280733e-04,
                       -9.998902e-01, 9.998621e-01, 7.523790e-03, 1.480755e-02]),
        "T": np.array([-4.069766e-03, -7.631618e-02, -2.717806e-01]),
    },
}


def synthetic_scan(n: int, seed: int = 0) -> np.ndarray:
    """(n, 4) float32 rows shaped like a velodyne sweep: a ring pattern around
    the sensor (all azimuths, so about half the points lie behind the camera),
    64 elevation rings, ranges 3-80 m."""
    g = np.random.default_rng(seed)
    az = g.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(g.integers(0, 64, n) * (26.8 / 63) - 24.8)
    r = g.uniform(3.0, 80.0, n)
    pts = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el),
                    g.uniform(0, 1, n)], axis=1)
    return pts.astype(np.float32)
