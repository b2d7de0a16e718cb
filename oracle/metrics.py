"""CPU restatement of the reference's evaluation metrics
(/root/reference/train.py:248-253 `normalize_image`, :311-336 `compute_errors`,
:338-343 `compute_depth_errors`).

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

**Parity unpinned**: `compute_errors` calls
`torchvision.transforms.functional.resize` (train.py:319-320), which is not
importable here; it is restated as bilinear interpolation with
align_corners=False and no antialiasing, the same assumption as in
`oracle/compute_loss.py`.  Everything else is plain torch and is restated with
the same operators.  The reference holds no fixture for these functions.

Generalised to a batch: sample b of a result equals the reference's batch-1
call on sample b.  The reference evaluates on the GPU, where the division of a
tensor by a host scalar (train.py:253) is a multiplication with the scalar's
float32 reciprocal; the CPU operator used here divides.  The two differ by at
most one unit in the last place, which the parity tolerance (1e-6) covers.
"""
import torch
import torch.nn.functional as F


def compute_depth_errors(gt, pred):
    """train.py:338-343, per sample: (B, ...) x2 -> (B,)."""
    rmse = (gt - pred) ** 2
    return torch.sqrt(rmse.flatten(1).mean(dim=1))


def compute_errors_one_side(gt, pred, height, width, maxdisp=128.0):
    """One side of train.py:311-336: gt (B,1,Ht,Wt), pred (B,H,W) -> rmse (B,)."""
    gt = gt.float()
    if tuple(gt.shape[-2:]) != (height, width):
        gt = F.interpolate(gt, (height, width), mode="bilinear", align_corners=False)   # :319
    gt = gt[:, 0]                                                                        # :321 (squeeze)
    pred = 1 / pred                                                                      # :324
    pred = torch.clamp(pred, min=1e-3, max=maxdisp)                                      # :327
    return compute_depth_errors(gt, pred.detach())                                       # :333


def normalize_image(x):
    """train.py:248-253 per sample: (B, ...) -> same shape."""
    out = torch.empty_like(x)
    for b in range(x.shape[0]):
        ma = float(x[b].max())
        mi = float(x[b].min())
        d = ma - mi if ma != mi else 1e5
        out[b] = (x[b] - mi) / d
    return out
