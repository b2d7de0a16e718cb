"""GPU parity of compute_loss (reference: train.py:272-309) against the CPU
restatement. Floating point, tolerance 1e-5 relative on the loss and each term."""
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _inputs(B, Ht, Wt, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    target = torch.rand(B, Ht, Wt, generator=g)
    pred = torch.rand(B, H, W, generator=g)
    return target, pred


@pytest.mark.parametrize("B,Ht,Wt,H,W", [(1, 375, 1242, 256, 512), (1, 375, 1242, 64, 128),
                                          (3, 192, 640, 192, 640), (2, 50, 70, 37, 45)])
def test_compute_loss_matches_oracle(B, Ht, Wt, H, W):
    from oracle.compute_loss import compute_loss_terms as ref_terms
    from depth_estimation_amd.loss import compute_loss_terms, compute_loss
    target, pred = _inputs(B, Ht, Wt, H, W, seed=H + B)
    ref = ref_terms(target, pred)
    got = compute_loss_terms(target.cuda(), pred.cuda())
    for k in ("loss", "ssim_term", "l1", "smooth"):
        r, g = ref[k], got[k].cpu()
        assert torch.allclose(g, r, rtol=RTOL, atol=0), (k, g, r)
    out = compute_loss(target.cuda(), pred.cuda())
    assert out.dim() == (0 if B == 1 else 1)


@pytest.mark.parametrize("B,Ht,Wt,H,W", [(1, 375, 1242, 256, 512), (2, 375, 1242, 192, 640), (2, 50, 70, 37, 45),
                                          (1, 40, 60, 64, 128)])
def test_antialiased_resize_matches_oracle(B, Ht, Wt, H, W):
    """torchvision's `resize` (train.py:276) anti-aliases tensors in some releases and not in
    others; both settings are compared with the oracle, at the reference's own 375x1242 -> 256x512,
    and the two settings must differ when the target is shrunk (the test is not vacuous)."""
    from oracle.compute_loss import compute_loss_terms as ref_terms
    from depth_estimation_amd.loss import compute_loss_terms
    target, pred = _inputs(B, Ht, Wt, H, W, seed=Ht + B)
    plain = compute_loss_terms(target.cuda(), pred.cuda(), antialias=False)
    for aa in (False, True):
        ref = ref_terms(target, pred, antialias=aa)
        got = compute_loss_terms(target.cuda(), pred.cuda(), antialias=aa)
        for k in ("loss", "ssim_term", "l1", "smooth"):
            assert torch.allclose(got[k].cpu(), ref[k], rtol=RTOL, atol=0), (aa, k, got[k], ref[k])
    if Ht > H:
        assert not torch.allclose(got["loss"], plain["loss"], rtol=1e-3, atol=0)


@pytest.mark.parametrize("B,Ht,Wt,H,W", [(1, 4, 4, 4, 4), (2, 9, 300, 9, 129), (1, 7, 5, 6, 260), (3, 5, 2, 11, 7),
                                          (1, 130, 127, 130, 127), (2, 64, 1, 64, 1 + 3)])
def test_edge_shapes_match_oracle(B, Ht, Wt, H, W):
    """Smallest sizes the reflect border allows, widths around the 128-column strip (a strip with a
    single live column, an odd width whose last pair straddles the edge), two-column sources."""
    from oracle.compute_loss import compute_loss_terms as ref_terms
    from depth_estimation_amd import _lib
    from depth_estimation_amd.loss import compute_loss_terms
    target, pred = _inputs(B, Ht, Wt, H, W, seed=3 * H + W)
    if Wt == 1:  # a one-column source cannot be resized by pairs of columns: refused, not mis-read
        with pytest.raises(_lib.DepthwarpError):
            compute_loss_terms(target.cuda(), pred.cuda())
        return
    for aa in (False, True):
        ref = ref_terms(target, pred, antialias=aa)
        got = compute_loss_terms(target.cuda(), pred.cuda(), antialias=aa)
        for k in ("loss", "ssim_term", "l1", "smooth"):
            assert torch.allclose(got[k].cpu(), ref[k], rtol=RTOL, atol=1e-7), (aa, k, got[k], ref[k])


def test_batched_equals_per_sample_loop():
    from depth_estimation_amd.loss import compute_loss
    target, pred = _inputs(4, 96, 320, 64, 128, seed=9)
    t, p = target.cuda(), pred.cuda()
    batched = compute_loss(t, p)
    single = torch.stack([compute_loss(t[i:i + 1], p[i:i + 1]) for i in range(4)])
    assert torch.equal(batched, single)


def test_large_batch_equals_per_sample_calls_to_rounding_and_is_reproducible():
    """At 64 x (375x1242 -> 256x512) the row chunks are taller than in a batch of one, so the
    partial sums are grouped differently: equal to rounding, and bitwise equal run to run."""
    from depth_estimation_amd.loss import compute_loss_terms
    target, pred = _inputs(64, 375, 1242, 256, 512, seed=21)
    t, p = target.cuda(), pred.cuda()
    a = compute_loss_terms(t, p)
    b = compute_loss_terms(t, p)
    for k in a:
        assert torch.equal(a[k], b[k])
    for i in (0, 17, 63):
        one = compute_loss_terms(t[i:i + 1], p[i:i + 1])
        for k in a:
            assert torch.allclose(a[k][i:i + 1], one[k], rtol=2e-6, atol=0), (i, k)


def test_float64_target_like_reference():
    from oracle.compute_loss import compute_loss as ref
    from depth_estimation_amd.loss import compute_loss
    target, pred = _inputs(1, 375, 1242, 256, 512, seed=1)
    got = compute_loss(target.double().cuda(), pred.cuda())
    assert torch.allclose(got.cpu(), ref(target.double(), pred), rtol=RTOL, atol=0)
    assert not got.requires_grad


@pytest.mark.parametrize("B,Ht,Wt,H,W", [(2, 375, 1242, 256, 512), (1, 64, 129, 64, 129), (2, 31, 45, 50, 70)])
def test_float64_target_is_read_in_place_and_equals_the_cast(B, Ht, Wt, H, W):
    """A float64 target goes to the kernel as it is (flag DW_CL_FLAG_TARGET_F64) and is rounded
    to float32 value by value on the way in: bit-identical to casting first, for every resize
    mode and for the metrics entry point."""
    from depth_estimation_amd import _lib
    from depth_estimation_amd.loss import compute_loss_terms
    target, pred = _inputs(B, Ht, Wt, H, W, seed=Ht)
    t64 = (target.double() + torch.rand(B, Ht, Wt, dtype=torch.float64) * 1e-9).cuda()   # not float32-representable
    p = pred.cuda()
    for aa in (False, True):
        a = compute_loss_terms(t64, p, antialias=aa)
        b = compute_loss_terms(t64.float(), p, antialias=aa)
        for k in a:
            assert torch.equal(a[k], b[k]), (aa, k)
    lib = _lib.lib()
    n = lib.dw_compute_loss_workspace_bytes(B, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    outs = []
    for t, flags in ((t64, _lib.DW_CL_FLAG_TARGET_F64), (t64.float(), 0)):
        out = torch.empty(B, 8, device="cuda")
        rc = lib.dw_compute_loss_metrics_fwd(B, Ht, Wt, H, W, t.data_ptr(), (p + 1e-3).data_ptr(), 1e-3, 128.0,
                                             out.data_ptr(), flags, ws.data_ptr(), n,
                                             torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(out[:, :7].clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Ht,Wt,H,W,f64,aa", [(375, 1242, 256, 512, True, False), (375, 1242, 256, 512, False, True),
                                               (192, 640, 192, 640, False, False), (50, 70, 37, 45, True, True)])
def test_left_right_pair_is_one_call_and_equals_two(Ht, Wt, H, W, f64, aa):
    """`compute_loss(l_depth, l_pred) + compute_loss(r_depth, r_pred)` (train.py:113, 177) through
    compute_loss_lr: one launch pair on the four tensors as they are.  Each view's terms are
    bit-identical to a call of its own (the row chunks are those of a one-sample call) and the
    sum is the float32 sum of the two losses; also against the CPU restatement."""
    from oracle.compute_loss import compute_loss as ref
    from depth_estimation_amd.loss import compute_loss, compute_loss_lr, compute_loss_terms
    lt, lp = _inputs(1, Ht, Wt, H, W, seed=5)
    rt, rp = _inputs(1, Ht, Wt, H, W, seed=6)
    if f64:
        lt, rt = lt.double(), rt.double()
    glt, glp, grt, grp = lt.cuda(), lp.cuda(), rt.cuda(), rp.cuda()
    total, terms = compute_loss_lr(glt, glp, grt, grp, antialias=aa, return_terms=True)
    assert total.dim() == 0 and not total.requires_grad
    for i, (t, p) in enumerate(((glt, glp), (grt, grp))):
        one = compute_loss_terms(t, p, antialias=aa)
        for j, k in enumerate(("loss", "ssim_term", "l1", "smooth")):
            assert torch.equal(terms[i, j], one[k][0]), (i, k)
    two = compute_loss(glt, glp, antialias=aa) + compute_loss(grt, grp, antialias=aa)
    assert torch.equal(total, two)
    assert torch.equal(compute_loss_lr(glt, glp, grt, grp, antialias=aa), total)
    want = ref(lt, lp, antialias=aa) + ref(rt, rp, antialias=aa)
    assert torch.allclose(total.cpu(), want.float(), rtol=RTOL, atol=0)
    # (1,1,H,W) maps, as the reference's views produce them, are accepted as they are
    assert torch.equal(compute_loss_lr(glt[None], glp[None], grt[None], grp[None], antialias=aa), total)


def test_left_right_pair_refuses_what_the_reference_refuses():
    from depth_estimation_amd import _lib
    from depth_estimation_amd.loss import compute_loss_lr
    a, b = torch.rand(1, 32, 32).cuda(), torch.rand(1, 32, 32).cuda()
    with pytest.raises(RuntimeError):
        compute_loss_lr(a, b, a, b, 64, 64)                       # not the configured size
    with pytest.raises(RuntimeError):
        compute_loss_lr(a, b, a, torch.rand(1, 16, 32).cuda())    # views of different sizes
    with pytest.raises(RuntimeError):
        compute_loss_lr(torch.rand(2, 32, 32).cuda(), b, a, b)    # a batch is not one map
    with pytest.raises(_lib.DepthwarpError):
        compute_loss_lr(a.cpu(), b.cpu(), a.cpu(), b.cpu())
    # the C entry point: a pair needs both buffers of the second view
    lib = _lib.lib()
    n = lib.dw_compute_loss_workspace_bytes(2, 32, 32)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = torch.empty(9, device="cuda")
    assert lib.dw_compute_loss_pair_fwd(32, 32, 32, 32, a.data_ptr(), b.data_ptr(), None, b.data_ptr(), out.data_ptr(),
                                        0, ws.data_ptr(), n, None) == -1  # DW_ERR_NULL


def test_shape_mismatch_raises():
    from depth_estimation_amd.loss import compute_loss
    with pytest.raises(RuntimeError):
        compute_loss(torch.rand(1, 32, 32).cuda(), torch.rand(1, 32, 32).cuda(), 64, 64)


def test_hip_matches_committed_golden_vectors():
    import os
    import numpy as np
    from depth_estimation_amd.loss import compute_loss_terms
    from tests.golden.make_golden import CL_CASES, cl_inputs
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "compute_loss.npz"))
    for i, c in enumerate(CL_CASES):
        t, p = cl_inputs(c)
        terms = compute_loss_terms(t.cuda(), p.cuda(), antialias=c.get("aa", False))
        got = np.stack([terms[k].cpu().numpy() for k in ("loss", "ssim_term", "l1", "smooth")], 1)
        assert np.allclose(got, gold[f"case{i}"], rtol=RTOL, atol=0), (i, got, gold[f"case{i}"])
