"""Reads the rare-path flag of the colour scatter after one forward pass: 1 = some taps left their patch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd.ops import PhotometricPlan
from depth_estimation_amd.synth import make_sequence
B, H, W = 12, 192, 640
seq = make_sequence(B, H, W, seed=0).to("cuda")
plan = PhotometricPlan(B, H, W, 4, "cuda", flags=4).bind(*seq.args())
plan.forward(True)
torch.cuda.synchronize()
gcol_bytes = (2 * B * 3 * H * W * 8 + 255) // 256 * 256
flag_off = plan.nbytes - gcol_bytes - 256
flag = plan.ws[flag_off:flag_off + 4].view(torch.int32)
gcol = plan.ws[plan.nbytes - gcol_bytes:].view(torch.int64)
print("spill flag", int(flag[0]), "non-zero rare-path entries", int((gcol != 0).sum()), "of", gcol.numel())
