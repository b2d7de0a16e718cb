"""The HIP kernels of the stages next to the headline path against the committed vectors of
`tests/golden/next_rows.npz` (soft-argmin forward and backward, cost volume, rmse and normalised
image, LiDAR projection): copies and integer / float64 selection bit for bit, floating point to
the bars of the parity tests."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hip_matches_committed_vectors_of_the_next_rows():
    from depth_estimation_amd.cost_volume import cost_volume
    from depth_estimation_amd.disparity import soft_argmin
    from depth_estimation_amd.lidar import project_scans
    from depth_estimation_amd.metrics import compute_loss_and_metrics, normalize_image
    from tests.golden.make_golden import next_inputs
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "next_rows.npz"))
    inp = next_inputs()
    # soft-argmin (model.py:417-429, 454-456), forward and backward
    cost = inp["cost"].cuda().requires_grad_(True)
    disp = soft_argmin(cost)
    w = torch.linspace(-1.0, 2.0, disp.numel()).view_as(disp).cuda()
    (disp * w).sum().backward()
    assert np.allclose(disp.detach().cpu().numpy(), gold["soft_argmin"], rtol=1e-5, atol=1e-6)
    g = torch.from_numpy(gold["soft_argmin_grad"])
    assert float((cost.grad.cpu() - g).abs().max() / g.abs().max()) <= 1e-4
    # cost volume (model.py:460-508): copies
    left, right = cost_volume(*(m.cuda() for m in inp["maps"]), maxdisp=5)
    assert np.array_equal(left.cpu().numpy(), gold["cost_left"])
    assert np.array_equal(right.cpu().numpy(), gold["cost_right"])
    # metrics (train.py:248-253, 311-343)
    m = compute_loss_and_metrics(inp["gt"].cuda(), inp["pred"].cuda())
    assert np.allclose(m["rmse"].cpu().numpy(), gold["rmse"], rtol=1e-5)
    assert np.allclose(normalize_image(inp["img"].cuda()).cpu().numpy(), gold["normalized"], rtol=1e-6, atol=1e-7)
    # LiDAR projection (load_dataset.py:116-148): float64, bit for bit
    scan = torch.from_numpy(inp["scan"])[None].cuda()
    pts, n_valid, depth = project_scans(scan, torch.tensor([scan.shape[1]], dtype=torch.int32),
                                        torch.from_numpy(inp["P"])[None], inp["shape"])
    n = int(n_valid[0])
    assert n == gold["lidar_pts"].shape[0]
    assert np.array_equal(pts[0, :n].cpu().numpy(), gold["lidar_pts"])
    assert np.array_equal(depth[0].cpu().numpy(), gold["lidar_depth"])
