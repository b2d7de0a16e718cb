import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd.ops import PhotometricPlan
from depth_estimation_amd.synth import make_sequence
seq = make_sequence(12, 192, 640, seed=0).to("cuda")
plan = PhotometricPlan(12, 192, 640, 4, "cuda").bind(*seq.args())
plan.forward(True)
plan.tile_kernel_only()
for _ in range(3): plan.forward(True)
torch.cuda.synchronize()
# gd0 is the last workspace section: B*H*W floats
ws = plan.ws.view(torch.float32)
gd0 = ws[-4*12*192*640:]
t = gd0[:40].cpu().view(4, 10)
names = ["loop head", "A", "barrier after A", "B", "G planes + barriers", "C", "pose partials + D", "loss sum", "prologue loads", "ystat + identity"]
print("per-wave cycles (rows=waves):")
print(t.int())
tot = t.sum(1)
print("total/wave", tot.int())
print("mean share:", {n: round(float(t[:, i].mean() / tot.mean()), 3) for i, n in enumerate(names)})
