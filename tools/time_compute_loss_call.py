import sys, time, torch
sys.path.insert(0, "/root/repo")
from depth_estimation_amd.loss import compute_loss
t64 = torch.rand(1, 375, 1242, dtype=torch.float64, device="cuda")
t32 = t64.float()
pred = torch.rand(1, 256, 512, device="cuda")
for name, t in (("f64 target", t64), ("f32 target", t32)):
    for _ in range(50): compute_loss(t, pred)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): compute_loss(t, pred)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "enqueue %.1f us/call, until drained %.1f us/call" % (1e6 * (t1 - t0) / 300, 1e6 * (t2 - t0) / 300))
