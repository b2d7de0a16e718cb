"""CPU restatement of the disparity regression at the end of
`DepthBlock.forward` (/root/reference/model/model.py).

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

The reference lines are plain torch calls, restated here with the same ATen
operators (the module itself cannot be imported: its file pulls in torchvision
and the dataset code, model.py:5-10):
    model.py:455-456   out.view(1, 2*maxdisp, H, W);  prob = F.softmax(-out, 1)
    model.py:418       squeeze
    model.py:421       F.softmax(-prob, dim=0)           (a second soft-max)
    model.py:424-425   d_grid = arange(2*maxdisp) repeated over H x W
    model.py:428       sum(softmax * d_grid, dim=0, keepdim=True)
The reference is batch 1; here the disparity axis is dim 1 of a (B, D, H, W)
volume and sample b equals the reference's result on sample b: the result is
(B, H, W), which at batch 1 is the reference's (1, H, W).  The reference
holds no fixture for this function: pinned only by using the same operators.
"""
import torch
import torch.nn.functional as F


def soft_argmin(cost: torch.Tensor) -> torch.Tensor:
    """cost (B, D, H, W) -> disparity (B, H, W)."""
    prob = F.softmax(-cost, 1)                       # model.py:456
    prob2 = F.softmax(-prob, 1)                      # model.py:421 (dim 0 after the squeeze)
    D = cost.shape[1]
    d_grid = torch.arange(D, dtype=cost.dtype).reshape(1, -1, 1, 1)   # model.py:424
    return torch.sum(prob2 * d_grid, dim=1)                           # model.py:428 (dim 0, keepdim, at batch 1)
