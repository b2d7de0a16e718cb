"""CPU restatement of `Dataset.generate_depth_map`
(/root/reference/dataset/load_dataset.py:99-153) from the projection matrix to
the sparse depth map, i.e. everything except file loading and the Delaunay
hole filling.

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

NumPy statements are restated with the same NumPy operators, with three
changes that do not alter values: `np.int` (removed from NumPy) is `int`; the
4-term dot products of line 122 are written as explicit multiply-adds in the
order ((p0 x + p1 y) + p2 z) + p3, because `np.dot` leaves the order to the
BLAS of the host (a test holds the two forms together to 1e-12); and the
O(duplicates x N) Python loop of lines 142-147 is grouped with `np.unique`
(same result: every group of points sharing a `sub2ind` key writes its minimum
depth at the pixel of its first point).  The `sub2ind` quirk (:196-200: the key
is row * (n - 1) + col - 1, which is not the pixel index) is kept.

The reference module cannot be imported (skimage, cv2, torchvision are absent)
and holds no fixture for this function: parity unpinned.
"""
import numpy as np


def velo_to_image_matrix(cam2cam: dict, velo2cam: dict, cam: int = 2) -> np.ndarray:
    """load_dataset.py:104-114: P_velo2im (3, 4) float64 from the two calibration dicts."""
    v2c = np.hstack((velo2cam['R'].reshape(3, 3), velo2cam['T'][..., np.newaxis]))
    v2c = np.vstack((v2c, np.array([0, 0, 0, 1.0])))
    R_cam2rect = np.eye(4)
    R_cam2rect[:3, :3] = cam2cam['R_rect_00'].reshape(3, 3)
    P_rect = cam2cam['P_rect_0' + str(cam)].reshape(3, 4)
    return np.dot(np.dot(P_rect, R_cam2rect), v2c)


def sub2ind(matrixSize, rowSub, colSub):
    m, n = matrixSize
    return rowSub * (n - 1) + colSub - 1                      # :196-200


def project_scan(points: np.ndarray, P: np.ndarray, im_shape, vel_depth=False, use_np_dot=False):
    """points (N, 4) float32 velodyne rows, P (3, 4) float64, im_shape (H, W).
    Returns (velo_pts_im (N', 3) float64, depth (H, W) float64) as they stand
    after load_dataset.py:134 and :148."""
    velo = points.copy()
    velo[:, 3] = 1.0                                          # :193
    velo = velo[velo[:, 0] >= 0, :]                           # :119
    if use_np_dot:
        velo_pts_im = np.dot(P, velo.T).T                     # :122
    else:
        v = velo.astype(np.float64)
        velo_pts_im = np.stack([((P[k, 0] * v[:, 0] + P[k, 1] * v[:, 1]) + P[k, 2] * v[:, 2]) + P[k, 3] * v[:, 3]
                                for k in range(3)], axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        velo_pts_im[:, :2] = velo_pts_im[:, :2] / velo_pts_im[:, 2][..., np.newaxis]   # :123
    if vel_depth:
        velo_pts_im[:, 2] = velo[:, 0]                        # :126
    velo_pts_im[:, 0] = np.round(velo_pts_im[:, 0]) - 1       # :130
    velo_pts_im[:, 1] = np.round(velo_pts_im[:, 1]) - 1
    val_inds = (velo_pts_im[:, 0] >= 0) & (velo_pts_im[:, 1] >= 0)                     # :132
    val_inds = val_inds & (velo_pts_im[:, 0] < im_shape[1]) & (velo_pts_im[:, 1] < im_shape[0])
    velo_pts_im = velo_pts_im[val_inds, :]

    depth = np.zeros((im_shape[:2]))                          # :137
    depth[velo_pts_im[:, 1].astype(int), velo_pts_im[:, 0].astype(int)] = velo_pts_im[:, 2]

    inds = sub2ind(depth.shape, velo_pts_im[:, 1], velo_pts_im[:, 0])                  # :141
    if len(inds):
        _, first, inverse, counts = np.unique(inds, return_index=True, return_inverse=True,
                                              return_counts=True)
        gmin = np.full(len(first), np.inf)
        np.minimum.at(gmin, inverse, velo_pts_im[:, 2])
        for g in np.nonzero(counts > 1)[0]:                   # :142-147
            x_loc = int(velo_pts_im[first[g], 0])
            y_loc = int(velo_pts_im[first[g], 1])
            depth[y_loc, x_loc] = gmin[g]
    depth[depth < 0] = 0                                      # :148
    return velo_pts_im, depth


def project_scan_loop(points, P, im_shape):
    """The duplicate handling exactly as the reference writes it (Counter +
    np.where per duplicate, :141-147); slow, for cross-checking small cases."""
    from collections import Counter
    velo_pts_im, _ = project_scan(points, P, im_shape)
    depth = np.zeros((im_shape[:2]))
    depth[velo_pts_im[:, 1].astype(int), velo_pts_im[:, 0].astype(int)] = velo_pts_im[:, 2]
    inds = sub2ind(depth.shape, velo_pts_im[:, 1], velo_pts_im[:, 0])
    dupe_inds = [item for item, count in Counter(inds).items() if count > 1]
    for dd in dupe_inds:
        pts = np.where(inds == dd)[0]
        x_loc = int(velo_pts_im[pts[0], 0])
        y_loc = int(velo_pts_im[pts[0], 1])
        depth[y_loc, x_loc] = velo_pts_im[pts, 2].min()
    depth[depth < 0] = 0
    return depth
