"""CPU checks of the oracles of the stages next to the headline path: closed
forms that follow from the reference lines they restate."""
import torch


def test_cost_volume_closed_form():
    from oracle.cost_volume import cost_volume
    B, F, h, w, D = 2, 3, 4, 8, 5
    imgl, imgr, carl, carr = [torch.arange(B * F * h * w, dtype=torch.float32).reshape(B, F, h, w) + 1000 * k
                              for k in range(4)]
    left, right = cost_volume(imgl, imgr, carl, carr, maxdisp=D)
    assert left.shape == right.shape == (B, D, 4 * F, h, w)
    # model.py:471-472: channel order at disparity 0
    assert torch.equal(left[:, 0], torch.cat([imgl, carl, imgr, carr], 1))
    assert torch.equal(right[:, 0], torch.cat([imgr, carr, imgl, carl], 1))
    for d in range(1, D):
        # model.py:497: [imgl, imgr shifted right by d, carl, carr shifted right by d]
        assert torch.equal(left[:, d, :F], imgl) and torch.equal(left[:, d, 2 * F:3 * F], carl)
        assert torch.equal(left[:, d, F:2 * F, :, d:], imgr[..., :w - d])
        assert torch.equal(left[:, d, 3 * F:, :, d:], carr[..., :w - d])
        assert float(left[:, d, F:2 * F, :, :d].abs().max()) == 0
        # model.py:498: [imgr, imgl shifted left by d, carr, carl shifted left by d]
        assert torch.equal(right[:, d, :F], imgr) and torch.equal(right[:, d, 2 * F:3 * F], carr)
        assert torch.equal(right[:, d, F:2 * F, :, :w - d], imgl[..., d:])
        assert torch.equal(right[:, d, 3 * F:, :, :w - d], carl[..., d:])
        assert float(right[:, d, 3 * F:, :, w - d:].abs().max()) == 0


def test_soft_argmin_closed_form():
    from oracle.soft_argmin import soft_argmin
    # constant cost: both soft-maxes are uniform, the result is the mean disparity index
    out = soft_argmin(torch.zeros(1, 8, 2, 3))
    assert out.shape == (1, 2, 3)   # the reference's (1, H, W), model.py:428
    assert torch.allclose(out, torch.full_like(out, 3.5))
    # the second soft-max of a normalised volume flattens even a one-hot first soft-max
    c = torch.full((1, 4, 1, 1), 100.0)
    c[0, 1] = -100.0
    p2 = torch.softmax(-torch.tensor([0.0, 1.0, 0.0, 0.0]), 0)
    want = float((p2 * torch.arange(4.0)).sum())
    assert abs(float(soft_argmin(c)) - want) < 1e-6


def test_metrics_closed_form():
    from oracle.metrics import compute_depth_errors, compute_errors_one_side, normalize_image
    gt = torch.full((2, 1, 4, 6), 3.0)
    pred = torch.full((2, 4, 6), 0.5)            # 1/pred = 2 -> error 1 everywhere
    pred[1] = 1e-6                               # 1/pred = 1e6 -> clamped to maxdisp
    r = compute_errors_one_side(gt, pred, 4, 6, maxdisp=128.0)
    assert torch.allclose(r, torch.tensor([1.0, 125.0]))
    assert torch.allclose(compute_depth_errors(torch.zeros(1, 5), torch.full((1, 5), 2.0)), torch.tensor([2.0]))
    x = torch.tensor([[1.0, 3.0, 2.0], [4.0, 4.0, 4.0]])
    n = normalize_image(x)
    assert torch.equal(n[0], torch.tensor([0.0, 1.0, 0.5])) and torch.equal(n[1], torch.zeros(3))


def test_lidar_oracle_forms_agree():
    import numpy as np
    from depth_estimation_amd.lidar import KITTI_LIKE_CALIB, synthetic_scan, velo_to_image_matrix
    from oracle import lidar as ol
    c = KITTI_LIKE_CALIB
    P = velo_to_image_matrix(c["cam2cam"], c["velo2cam"], 2)
    assert np.allclose(P, ol.velo_to_image_matrix(c["cam2cam"], c["velo2cam"], 2), rtol=0, atol=0)
    scan = synthetic_scan(3000, seed=0)
    shape = (94, 311)
    Ps = np.diag([shape[1] / 1242.0, shape[0] / 375.0, 1.0]) @ P
    pts, depth = ol.project_scan(scan, Ps, shape)
    # explicit multiply-add order vs np.dot (host BLAS): same up to rounding of the last place
    pts_dot, _ = ol.project_scan(scan, Ps, shape, use_np_dot=True)
    assert pts.shape == pts_dot.shape and np.allclose(pts, pts_dot, rtol=1e-12, atol=1e-12)
    # np.unique grouping vs the literal Counter / np.where loop of the reference
    assert np.array_equal(depth, ol.project_scan_loop(scan, Ps, shape))
    assert len(pts) > 200 and (depth > 0).sum() > 100
    # every kept point is inside the image and in front of the sensor
    assert pts[:, 0].min() >= 0 and pts[:, 0].max() < shape[1] and pts[:, 1].max() < shape[0]


def test_oracles_reproduce_the_committed_vectors():
    """`tests/golden/next_rows.npz` (written by tests/golden/make_golden.py from these oracles on
    seeded inputs): a change to a restatement that alters its results shows up here."""
    import os
    import numpy as np
    from tests.golden.make_golden import next_inputs, next_outputs
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "next_rows.npz"))
    got = next_outputs(next_inputs())
    assert sorted(gold.files) == sorted(got)
    for k in ("cost_left", "cost_right", "lidar_pts", "lidar_depth"):   # copies and float64 selection: exact
        assert np.array_equal(got[k], gold[k]), k
    for k in ("soft_argmin", "soft_argmin_grad", "rmse", "normalized"):
        assert np.allclose(got[k], gold[k], rtol=1e-6, atol=1e-7), k
    assert (gold["lidar_depth"] > 0).sum() > 100 and gold["lidar_pts"].shape[0] > 200
