"""CPU oracle for the view-synthesis / photometric-loss path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package
(`depth_estimation_amd/`) may import from here: only `tests/`, the smoke check
in `__graft_entry__.py` and the `cpu_baseline` leg of `bench.py` do, and there
only as the checker.

Parity status
-------------
* `oracle.compute_loss` restates `/root/reference/train.py:272-309`.  Its
  arithmetic lives in kornia and torchvision, neither of which is vendored,
  pinned or importable in the build container, and the reference holds no test,
  fixture or golden vector for it: **parity unpinned**.
* `oracle.photometric` restates stages the reference does not contain at all
  (backprojection, projection, warp, SSIM+L1 reprojection error, minimum
  reprojection, auto-mask, edge-aware smoothness).  The definition followed is
  the public one (Godard et al., ICCV 2019); `oracle.aten_ref` states it with
  stock ATen operators and `tests/test_oracle.py` holds the two to each other.
  **Parity unpinned** against the reference for the same reason.
"""
