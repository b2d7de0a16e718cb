"""The allocation-free entry point dw_compute_loss_fwd_static (fixed scratch array inside the
library) against dw_compute_loss_fwd (caller workspace): same kernels, so the results must be
bit-identical -- on two calls in a row, the second with a smaller input, which also shows that
what the first call left in the shared array does not leak into the second."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _with_workspace(lib, target, pred, flags):
    B, _, H, W = pred.shape
    n = lib.dw_compute_loss_workspace_bytes(B, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = torch.full((B, 4), float("nan"), device="cuda")
    rc = lib.dw_compute_loss_fwd(B, target.shape[2], target.shape[3], H, W, target.data_ptr(), pred.data_ptr(),
                                 out.data_ptr(), flags, ws.data_ptr(), n,
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out


def _static(lib, target, pred, flags):
    B, _, H, W = pred.shape
    out = torch.full((B, 4), float("nan"), device="cuda")
    rc = lib.dw_compute_loss_fwd_static(B, target.shape[2], target.shape[3], H, W, target.data_ptr(),
                                        pred.data_ptr(), out.data_ptr(), flags,
                                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.dw_strerror(rc)
    return out


def test_static_scratch_entry_point_matches_the_workspace_one_twice_in_a_row():
    from depth_estimation_amd import _lib
    from oracle.compute_loss import compute_loss_terms as ref_terms
    lib = _lib.lib()
    g = torch.Generator().manual_seed(5)
    cases = [((3, 1, 375, 1242), (3, 1, 256, 512), 0),      # the reference's own sizes, batched
             ((1, 1, 50, 70), (1, 1, 37, 45), _lib.DW_CL_FLAG_ANTIALIAS)]   # then a smaller problem
    for tshape, pshape, flags in cases:
        target = torch.rand(*tshape, generator=g).cuda()
        pred = torch.rand(*pshape, generator=g).cuda()
        want = _with_workspace(lib, target, pred, flags)
        got = _static(lib, target, pred, flags)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all()
        assert torch.equal(got, want), (got, want)
        # and both agree with the CPU restatement (1e-5 relative)
        ref = ref_terms(target.cpu(), pred.cpu(), antialias=bool(flags))
        for i, k in enumerate(("loss", "ssim_term", "l1", "smooth")):
            assert torch.allclose(got[:, i].cpu(), ref[k], rtol=1e-5, atol=0), k


def test_static_scratch_entry_point_refuses_what_does_not_fit_and_python_wrapper():
    from depth_estimation_amd import _lib
    from depth_estimation_amd.loss import compute_loss_into, compute_loss_terms
    lib = _lib.lib()
    # 8 floats per 128-column strip, chunk of at most 64 rows and sample: 64 x 4096 x 2048 needs
    # 2 MiB of partial sums
    B, H, W = 64, 4096, 2048
    assert 32 * (W // 128) * (H // 64) * B > _lib.DW_STATIC_SCRATCH_BYTES
    x = torch.zeros(1, device="cuda")
    rc = lib.dw_compute_loss_fwd_static(B, H, W, H, W, x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, None)
    assert rc == -6 and b"too large" in lib.dw_strerror(rc)
    assert lib.dw_compute_loss_fwd_static(1, 8, 8, 8, 8, None, None, None, 0, None) == -1
    assert lib.dw_compute_loss_fwd_static(1, 8, 8, 8, 8, x.data_ptr(), x.data_ptr(), x.data_ptr(), 8, None) == -5
    g = torch.Generator().manual_seed(1)
    target, pred = torch.rand(2, 1, 96, 320, generator=g).cuda(), torch.rand(2, 1, 64, 128, generator=g).cuda()
    out = torch.empty(2, 4, device="cuda")
    compute_loss_into(out, target, pred)
    want = compute_loss_terms(target, pred)
    assert torch.equal(out[:, 0], want["loss"]) and torch.equal(out[:, 3], want["smooth"])
    # the wrapper raises on a problem beyond the scratch array: here through the batch size alone
    # (one strip, one chunk, 32 bytes per sample: more than 32768 samples), 10 MB of inputs
    Bb = 40000
    assert 32 * Bb > _lib.DW_STATIC_SCRATCH_BYTES
    with pytest.raises(_lib.DepthwarpError):
        compute_loss_into(torch.empty(Bb, 4, device="cuda"), torch.zeros(Bb, 1, 8, 8, device="cuda"),
                          torch.zeros(Bb, 1, 8, 8, device="cuda"))


def test_static_scratch_entry_point_allocates_nothing_and_runs_under_graph_capture():
    """No device memory is taken by the call (free memory of the device and the torch allocator's
    books are the same before and after), and it can be recorded into a HIP graph -- where an
    allocation would end the capture with an error -- and replayed on new inputs."""
    from depth_estimation_amd.loss import compute_loss_into, compute_loss_terms
    g = torch.Generator().manual_seed(9)
    target, pred = torch.rand(2, 1, 375, 1242, generator=g).cuda(), torch.rand(2, 1, 256, 512, generator=g).cuda()
    out = torch.empty(2, 4, device="cuda")
    compute_loss_into(out, target, pred)            # first call on the device: outside capture
    torch.cuda.synchronize()
    free0, booked0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
    for _ in range(3):
        compute_loss_into(out, target, pred)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0 and torch.cuda.memory_allocated() == booked0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        compute_loss_into(out, target, pred)
    target.copy_(torch.rand(2, 1, 375, 1242, generator=g))
    pred.copy_(torch.rand(2, 1, 256, 512, generator=g))
    out.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    want = compute_loss_terms(target, pred)
    for i, k in enumerate(("loss", "ssim_term", "l1", "smooth")):
        assert torch.equal(out[:, i], want[k]), k
