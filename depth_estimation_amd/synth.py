"""Seeded synthetic 3-frame sequences for tests and benchmarks.

There is no dataset in reach of the build, so every measurement and parity
test runs on inputs made here: smooth random colour images, source frames that
are shifted copies of the target plus noise, sigmoid-of-noise disparities at
four scales, the KITTI-style intrinsics the reference carries at
`/root/reference/dataset/load_dataset.py:21-24` (normalised, scaled to pixels
here), and small random rigid motions.  Everything is generated on the CPU from
one `torch.Generator` so that a GPU run and a CPU run see identical bits, and
only from uniform random numbers and correctly rounded IEEE operations (add,
mul, div, sqrt, abs): no libm or vectorised transcendental is involved, so two
hosts with different CPUs produce the same inputs and the committed golden
outputs stay valid on both.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F

# Normalised pinhole intrinsics (fx, fy, cx, cy as fractions of width/height).
K_NORM = ((0.58, 0.0, 0.5, 0.0),
          (0.0, 1.92, 0.5, 0.0),
          (0.0, 0.0, 1.0, 0.0),
          (0.0, 0.0, 0.0, 1.0))


@dataclass
class Sequence:
    disp: List[torch.Tensor]        # S x (B,1,H/2^s,W/2^s)
    color: List[torch.Tensor]       # [target, src(-1), src(+1)], (B,3,H,W)
    color_pyr: List[torch.Tensor]   # S x (B,3,H/2^s,W/2^s); [0] is color[0]
    K: torch.Tensor                 # (B,4,4)
    inv_K: torch.Tensor             # (B,4,4)
    T: torch.Tensor                 # (B,2,4,4)
    tie_noise: Optional[torch.Tensor] = None  # (B,2,H,W)

    def to(self, device, dtype=None):
        def mv(t):
            return None if t is None else t.to(device=device, dtype=dtype or t.dtype).contiguous()
        return Sequence([mv(d) for d in self.disp], [mv(c) for c in self.color],
                        [mv(c) for c in self.color_pyr], mv(self.K), mv(self.inv_K),
                        mv(self.T), mv(self.tie_noise))

    def args(self):
        return (self.disp, self.color, self.color_pyr, self.K, self.inv_K, self.T,
                self.tie_noise)


def intrinsics(B: int, H: int, W: int, dtype=torch.float32):
    K = torch.tensor(K_NORM, dtype=torch.float64)
    K[0] *= W
    K[1] *= H
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    inv_K = torch.eye(4, dtype=torch.float64)
    inv_K[0, 0], inv_K[0, 2] = 1.0 / fx, -cx / fx
    inv_K[1, 1], inv_K[1, 2] = 1.0 / fy, -cy / fy
    return (K.to(dtype)[None].repeat(B, 1, 1).contiguous(),
            inv_K.to(dtype)[None].repeat(B, 1, 1).contiguous())


def axis_angle_to_matrix(vec: torch.Tensor) -> torch.Tensor:
    """Rodrigues formula, (B,3) -> (B,3,3).  Differentiable."""
    angle = vec.norm(dim=1, keepdim=True).clamp_min(1e-12)
    axis = vec / angle
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = axis[:, 0:1], axis[:, 1:2], axis[:, 2:3]
    rows = [torch.cat([x * x * C + ca, x * y * C - z * sa, x * z * C + y * sa], 1),
            torch.cat([x * y * C + z * sa, y * y * C + ca, y * z * C - x * sa], 1),
            torch.cat([x * z * C - y * sa, y * z * C + x * sa, z * z * C + ca], 1)]
    return torch.stack(rows, 1)


def pose_to_matrix(axisangle: torch.Tensor, translation: torch.Tensor,
                   invert: bool = False) -> torch.Tensor:
    """(B,3), (B,3) -> (B,4,4) rigid transform  T = [R | t]  (or its inverse)."""
    R = axis_angle_to_matrix(axisangle)
    t = translation
    if invert:
        R = R.transpose(1, 2)
        t = -torch.bmm(R, t[:, :, None])[:, :, 0]
    B = R.shape[0]
    top = torch.cat([R, t[:, :, None]], 2)
    bottom = torch.zeros(B, 1, 4, dtype=R.dtype, device=R.device)
    bottom[:, 0, 3] = 1
    return torch.cat([top, bottom], 1)


def _box(x, k):
    """k x k mean with replicate border, as explicit shifted adds (fixed order)."""
    pad = k // 2
    p = F.pad(x, (pad, pad, pad, pad), mode="replicate")
    H, W = x.shape[-2:]
    acc = torch.zeros_like(x)
    for dy in range(k):
        for dx in range(k):
            acc = acc + p[..., dy:dy + H, dx:dx + W]
    return acc / float(k * k)


def _down2(x):
    return ((x[..., 0::2, 0::2] + x[..., 0::2, 1::2]) + (x[..., 1::2, 0::2] + x[..., 1::2, 1::2])) / 4.0


def _small_rotation(vec: torch.Tensor) -> torch.Tensor:
    """Rodrigues formula with the sine / cosine of the (small) angle taken from
    their Taylor polynomials, so that only exact IEEE operations are used."""
    a2 = (vec * vec).sum(1, keepdim=True)
    sinc = 1 - a2 / 6 + a2 * a2 / 120          # sin(a) / a
    cosc = 0.5 - a2 / 24 + a2 * a2 / 720       # (1 - cos(a)) / a^2
    x, y, z = vec[:, 0:1], vec[:, 1:2], vec[:, 2:3]
    ca = 1 - a2 * cosc
    rows = [torch.cat([x * x * cosc + ca, x * y * cosc - z * sinc, x * z * cosc + y * sinc], 1),
            torch.cat([x * y * cosc + z * sinc, y * y * cosc + ca, y * z * cosc - x * sinc], 1),
            torch.cat([x * z * cosc - y * sinc, y * z * cosc + x * sinc, z * z * cosc + ca], 1)]
    return torch.stack(rows, 1)


def make_sequence(B: int, H: int, W: int, S: int = 4, seed: int = 0,
                  tie_noise: bool = False, dtype=torch.float32) -> Sequence:
    assert H % (1 << (S - 1)) == 0 and W % (1 << (S - 1)) == 0
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *shape: torch.rand(*shape, generator=g, dtype=torch.float64)

    def rndn(*shape):  # ~N(0,1): centred sum of four uniforms, exact operations only
        u = ((rnd(*shape) + rnd(*shape)) + (rnd(*shape) + rnd(*shape))) - 2.0
        return u * 1.7320508075688772

    def squash(x):  # (0,1)-valued, sigmoid-like, rational
        return 0.5 + 0.5 * x / (1.0 + x.abs())

    tgt = _box(rnd(B, 3, H, W), 5)
    lo = tgt.amin(dim=(1, 2, 3), keepdim=True)
    hi = tgt.amax(dim=(1, 2, 3), keepdim=True)
    tgt = (tgt - lo) / (hi - lo)
    color = [tgt]
    shifts = []
    for f in range(2):
        dx = int(torch.randint(1, 4, (1,), generator=g)) * (1 if f else -1)
        dy = int(torch.randint(-1, 2, (1,), generator=g))
        shifts.append((dx, dy))
        src = torch.roll(tgt, shifts=(dy, dx), dims=(2, 3)) + 0.02 * rndn(B, 3, H, W)
        color.append(src.clamp(0, 1))
    pyr = [color[0]]
    for s in range(1, S):
        pyr.append(_down2(pyr[-1]))

    disp = []
    for s in range(S):
        h, w = H >> s, W >> s
        disp.append(0.02 + 0.06 * _box(squash(rndn(B, 1, h, w)), 3))

    K, inv_K = intrinsics(B, H, W, dtype=torch.float64)
    Ts = []
    # Depth is ~2-4 m for the disparities above; a translation of
    # shift * depth / f moves the re-projection by about the shift the source
    # frame was rolled by, so warped and identity errors compete per pixel.
    fx, fy = float(K[0, 0, 0]), float(K[0, 1, 1])
    for f in range(2):
        dx, dy = shifts[f]
        aa = 0.003 * rndn(B, 3)
        tr = 0.004 * rndn(B, 3)
        tr[:, 0] += -dx * 3.0 / fx
        tr[:, 1] += -dy * 3.0 / fy
        Tf = torch.zeros(B, 4, 4, dtype=torch.float64)
        Tf[:, :3, :3] = _small_rotation(aa)
        Tf[:, :3, 3] = tr
        Tf[:, 3, 3] = 1.0
        Ts.append(Tf)
    T = torch.stack(Ts, 1)
    noise = 1e-5 * rndn(B, 2, H, W) if tie_noise else None

    seq = Sequence(disp, color, pyr, K, inv_K, T, noise)
    return seq.to("cpu", dtype)


def algorithmic_bytes(B: int, H: int, W: int, S: int = 4, F_: int = 2) -> dict:
    """Compulsory HBM traffic of the fully fused op: every distinct input read
    once and every output written once per pass (fp32 maps, u8 index masks)."""
    N = H * W
    p = sum(4.0 ** -s for s in range(S))
    fwd = (1 + F_) * 3 * 4 * N + 4 * p * N + 12 * (p - 1) * N + S * N
    bwd = ((1 + F_) * 3 * 4 * N + 4 * p * N + 12 * (p - 1) * N) + S * N + 4 * p * N
    return {"fwd": B * fwd, "bwd": B * bwd, "total": B * (fwd + bwd)}
