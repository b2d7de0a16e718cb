"""CPU restatement of `DepthBlock.cost_volume`
(/root/reference/model/model.py:460-508).

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

The reference lines are plain torch calls (zeros / cat / slicing / stack),
restated here with the same operators; the module itself cannot be imported
(its file pulls in torchvision and the dataset code, model.py:5-10).  Two
things are generalised, nothing else: the zero block takes its batch and
height from the inputs instead of `Config.batch_size` / `Config.height // 2`
(model.py:478), and it lives on the inputs' device instead of `.cuda()`.

Kept as in the reference, including the quirk that the channel order at
disparity 0 differs from the order at every other disparity:
    d == 0 (model.py:471-472)  left  = [imgl, carl, imgr, carr]
                               right = [imgr, carr, imgl, carl]
    d  > 0 (model.py:497-498)  left  = [imgl, shift_right(imgr, d), carl, shift_right(carr, d)]
                               right = [imgr, shift_left(imgl, d),  carr, shift_left(carl, d)]
with zero fill (model.py:478-494).  The operation only copies values, so the
restatement is exact; the reference holds no fixture for it.
"""
import torch


def cost_volume(imgl, imgr, carl, carr, maxdisp=64):
    """Four (B, F, h, w) feature maps -> two (B, maxdisp, 4F, h, w) volumes.

    `maxdisp` is the reference's `self.maxdisp = Config.maxdisp // 2` (model.py:340).
    """
    B, Fs, h, w = imgl.shape
    left_list, right_list = [], []
    for d in range(maxdisp):                                                  # model.py:467
        if d == 0:
            leftplusright = torch.cat([imgl, carl, imgr, carr], dim=1)       # :471
            rightplusleft = torch.cat([imgr, carr, imgl, carl], dim=1)       # :472
        else:
            zero = torch.zeros(B, Fs, h, d, dtype=imgl.dtype, device=imgl.device)   # :478
            rightmove = torch.cat([zero, imgr], dim=3)[:, :, :, :-d]         # :481, :489
            leftmove = torch.cat([imgl, zero], dim=3)[:, :, :, d:]           # :482, :490
            rightmove1 = torch.cat([zero, carr], dim=3)[:, :, :, :-d]        # :485, :493
            leftmove1 = torch.cat([carl, zero], dim=3)[:, :, :, d:]          # :486, :494
            leftplusright = torch.cat([imgl, rightmove, carl, rightmove1], dim=1)   # :497
            rightplusleft = torch.cat([imgr, leftmove, carr, leftmove1], dim=1)     # :498
        left_list.append(leftplusright)
        right_list.append(rightplusleft)
    return torch.stack(left_list, dim=1), torch.stack(right_list, dim=1)      # :504-505
