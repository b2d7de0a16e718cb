"""CPU restatement of `Train.compute_loss` (/root/reference/train.py:272-309).

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

**Parity unpinned.**  The reference calls three third-party routines that are
neither vendored nor version-pinned nor importable in the build container, and
it holds no test or fixture for this function:
  * `torchvision.transforms.functional.resize`   (train.py:276)
  * `kornia.filters.spatial_gradient`            (train.py:284, 288)
  * `kornia.losses.ssim_loss(window_size=7)`     (train.py:302)
Their semantics are restated below from their published definitions
(kornia 0.6-0.7, torchvision 0.13-0.16, contemporaneous with the reference's use
of `Image.ANTIALIAS` and `np.int`); these are assumptions, recorded here:
  * resize      bilinear, align_corners=False; `antialias` is a flag (False is
                the tensor default of those torchvision versions).
  * Sobel       kernels [[-1,0,1],[-2,0,2],[-1,0,1]] and its transpose, divided
                by 8 (normalized=True), replicate padding, cross-correlation;
                output index 0 = d/dx, 1 = d/dy.
  * SSIM        7x7 Gaussian window, sigma 1.5, normalised, reflect padding;
                C1 = 0.01^2, C2 = 0.03^2 (max_val 1);
                ssim = num / (den + 1e-12); loss map clamp((1-ssim)/2, 0, 1); mean.

Quirks of the reference that are kept on purpose:
  * the whole body runs under `no_grad` (train.py:274): forward only;
  * `pred_edges` is computed and dropped; the "pred" gradients are the TARGET's
    Sobel responses (train.py:288-290);
  * the edge weights are scalars exp(+mean|grad|), not per-pixel exp(-|grad|)
    (train.py:292-293);
  * the SSIM term is  1 - ssim_loss  (train.py:302);
  * the mixing weights 0.75 / 0.20 / 0.15 sum to 1.10 (train.py:307).
The reference feeds a float64 target (NumPy pipeline) and a float32 prediction;
here the target is cast to float32 first.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gaussian_window(size: int = 7, sigma: float = 1.5, dtype=torch.float32):
    x = torch.arange(size, dtype=dtype) - size // 2
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def spatial_gradient(x):
    """(B,C,H,W) -> (B,C,2,H,W)."""
    B, C, H, W = x.shape
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype) / 8.0
    k = torch.stack([kx, kx.t()])[:, None]
    p = F.pad(x.reshape(B * C, 1, H, W), (1, 1, 1, 1), mode="replicate")
    return F.conv2d(p, k).reshape(B, C, 2, H, W)


def _filter7(x, win2d):
    p = F.pad(x, (3, 3, 3, 3), mode="reflect")
    return F.conv2d(p, win2d[None, None])


def ssim_loss7(pred, target, eps=1e-12):
    """kornia `ssim_loss(img1=pred, img2=target, 7, reduction='mean')`, per
    sample: (B,1,H,W) x2 -> (B,)."""
    g = gaussian_window(7, 1.5, pred.dtype)
    win = g[:, None] * g[None, :]
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = _filter7(pred, win), _filter7(target, win)
    s11 = _filter7(pred * pred, win) - mu1 * mu1
    s22 = _filter7(target * target, win) - mu2 * mu2
    s12 = _filter7(pred * target, win) - mu1 * mu2
    num = (2 * mu1 * mu2 + C1) * (2 * s12 + C2)
    den = (mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2)
    ssim = num / (den + eps)
    return torch.clamp((1 - ssim) / 2, 0, 1).mean(dim=(1, 2, 3))


def compute_loss_terms(target, pred, height=None, width=None, antialias=False):
    """Per-sample terms.  target (B,1,Ht,Wt) or (B,Ht,Wt); pred (B,1,H,W) or
    (B,H,W).  Returns dict of (B,) tensors: loss, ssim_term, l1, smooth."""
    with torch.no_grad():
        if pred.dim() == 3:
            pred = pred[:, None]
        if target.dim() == 3:
            target = target[:, None]
        pred = pred.float()
        target = target.float()
        H = height or pred.shape[-2]
        W = width or pred.shape[-1]
        if target.shape[-2:] != (H, W):
            target = F.interpolate(target, (H, W), mode="bilinear", align_corners=False,
                                   antialias=antialias)
        edges = spatial_gradient(target)                       # train.py:284
        dx_true, dy_true = edges[:, :, 0], edges[:, :, 1]
        dx_pred, dy_pred = dx_true, dy_true                    # train.py:289-290
        wx = torch.exp(dx_true.abs().mean(dim=(1, 2, 3)))      # train.py:292
        wy = torch.exp(dy_true.abs().mean(dim=(1, 2, 3)))
        smooth = ((dx_pred * wx[:, None, None, None]).abs().mean(dim=(1, 2, 3))
                  + (dy_pred * wy[:, None, None, None]).abs().mean(dim=(1, 2, 3)))
        ssim_term = 1 - ssim_loss7(pred, target)               # train.py:302
        l1 = (target - pred).abs().mean(dim=(1, 2, 3))         # train.py:305
        loss = 0.75 * ssim_term + 0.20 * l1 + 0.15 * smooth    # train.py:307
    return {"loss": loss, "ssim_term": ssim_term, "l1": l1, "smooth": smooth}


def compute_loss(target, pred, height=None, width=None, antialias=False):
    """Reference call shape: target (1,Ht,Wt), pred (1,H,W) -> 0-dim tensor.
    Batched inputs give the (B,) vector of the per-sample results."""
    out = compute_loss_terms(target, pred, height, width, antialias)["loss"]
    return out[0] if out.numel() == 1 else out
