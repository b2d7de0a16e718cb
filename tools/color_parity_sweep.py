#!/usr/bin/env python3
"""Gradient with respect to the warped source images (DW_FLAG_COLOR_GRAD) against autograd of the CPU
oracle over many seeds and a few shapes, with the poses of tools/parity_sweep.py (odd seeds: translations
that grow with the seed, so that taps leave their tile's patch and take the rare path): 1e-4 relative
(max-norm) per frame, the same bits on a second run, and the loss and masks of the plan without the flag.
usage: color_parity_sweep.py [seeds] [summary.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd._lib import DW_FLAG_COLOR_GRAD  # noqa: E402
from depth_estimation_amd.ops import PhotometricPlan  # noqa: E402
from tests.test_photometric_gpu import _oracle_color_grad, sweep_case  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = ((2, 32, 96), (1, 48, 160), (1, 40, 104), (1, 96, 320))
worst, bad, spilled, cases = 0.0, [], 0, 0
for seed in range(n):
    for B, H, W in SHAPES:
        seq = sweep_case(seed, B, H, W)
        want = _oracle_color_grad(seq)
        g = seq.to("cuda")
        plain = PhotometricPlan(B, H, W, 4, "cuda").bind(*g.args())
        plain.forward(True)
        plan = PhotometricPlan(B, H, W, 4, "cuda", flags=DW_FLAG_COLOR_GRAD).bind(*g.args())
        got = []
        for _ in range(2):
            plan.forward(True)
            got.append([c.clone() for c in plan.backward_color()])
        torch.cuda.synchronize()
        cases += 1
        ok = torch.equal(plan.loss, plain.loss) and torch.equal(plan.mask, plain.mask)
        errs = []
        for f in range(2):
            scale = float(want[f].abs().max())
            err = float((got[0][f].cpu() - want[f]).abs().max()) / scale if scale > 0 else float(got[0][f].abs().max())
            errs.append(err)
            ok = ok and err <= 1e-4 and torch.equal(got[0][f], got[1][f])
        worst = max(worst, *errs)
        if not ok:
            bad.append({"seed": seed, "shape": [B, H, W], "errs": errs})
            print("MISMATCH", bad[-1], flush=True)
    if seed % 10 == 9:
        print(f"seed {seed}: {cases} cases, worst {worst:.2e}, {len(bad)} outside the bar", flush=True)
summary = {"cases": cases, "seeds": n, "shapes": [list(s) for s in SHAPES], "worst_rel_err": worst,
           "outside_1e-4_or_not_reproducible": bad}
print(json.dumps(summary))
if len(sys.argv) > 2:
    json.dump(summary, open(sys.argv[2], "w"), indent=1)
sys.exit(1 if bad else 0)
