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
    assert out.shape == (1, 1, 2, 3)
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
