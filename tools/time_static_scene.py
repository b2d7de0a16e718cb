"""Tile-kernel time on the bench workload and on a static scene (source frames equal to the target:
the auto-mask removes every pixel, so no wave has a re-projection winner)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd.ops import PhotometricPlan
from depth_estimation_amd.synth import make_sequence

B, H, W = 12, 192, 640
for name in ("bench", "static", "half-static"):
    seq = make_sequence(B, H, W, seed=0)
    if name == "static":
        seq.color = [seq.color[0], seq.color[0].clone(), seq.color[0].clone()]
    if name == "half-static":
        c = [seq.color[0], seq.color[1].clone(), seq.color[2].clone()]
        for f in (1, 2):
            c[f][..., : H // 2, :] = seq.color[0][..., : H // 2, :]
        seq.color = c
    seq = seq.to("cuda")
    plan = PhotometricPlan(B, H, W, 4, "cuda").bind(*seq.args())
    plan.forward(True)
    plan.tile_kernel_only()
    for _ in range(5):
        plan.forward(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        plan.forward(True)
    e1.record()
    torch.cuda.synchronize()
    won = float(((plan.mask & 3) >= 2).float().mean())
    print(f"{name:12s} tile kernel {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us   re-projection wins at {100 * won:.1f} % of the pixels")
