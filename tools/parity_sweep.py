#!/usr/bin/env python3
"""HIP path against the CPU oracle over many seeds and a few shapes, with poses from mild to
extreme: masks must be equal bit for bit, loss within 1e-5, gradients within 1e-4 (max-norm;
disparity gradients outside the smoothness ties, see tests/test_photometric_gpu.py).
usage: parity_sweep.py [seeds] [summary.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd import ops  # noqa: E402
from depth_estimation_amd.synth import make_sequence  # noqa: E402
from oracle import photometric as ph  # noqa: E402
from tests.test_photometric_gpu import smoothness_ties, sweep_case  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
failures = []
n_ties = 0
worst = [0.0, 0.0, 0.0]
for seed in range(n):
    for B, H, W in ((2, 32, 96), (1, 48, 160), (1, 40, 104)):
        seq = sweep_case(seed, B, H, W)
        disp = [d.clone().requires_grad_(True) for d in seq.disp]
        T = seq.T.clone().requires_grad_(True)
        loss, aux = ph.photometric_stack(disp, seq.color, seq.color_pyr, seq.K, seq.inv_K, T, seq.tie_noise,
                                         return_aux=True)
        loss.backward()
        masks = torch.stack([ph.pack_mask(aux["idx"][s], aux["valid"][s]) for s in range(4)])
        gq = seq.to("cuda")
        hd = [d.clone().requires_grad_(True) for d in gq.disp]
        hT = gq.T.clone().requires_grad_(True)
        terms, mask = ops.photometric_forward(hd, gq.color, gq.color_pyr, gq.K, gq.inv_K, hT, gq.tie_noise)
        terms[0].backward()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        e_loss = abs(float(terms[0]) - float(loss)) / abs(float(loss))
        ties = [smoothness_ties(d) for d in seq.disp]
        n_ties += sum(int(t.sum()) for t in ties)
        e_gd = max(float(((h.grad.cpu() - d.grad).abs() * (~t)).max() / d.grad.abs().max())
                   for h, d, t in zip(hd, disp, ties))
        e_gT = rel(hT.grad.cpu()[:, :, :3], T.grad[:, :, :3]) if float(T.grad.abs().max()) > 0 else 0.0
        nm = int((mask.cpu() != masks).sum())
        worst = [max(worst[0], e_loss), max(worst[1], e_gd), max(worst[2], e_gT)]
        if nm or e_loss > 1e-5 or e_gd > 1e-4 or e_gT > 1e-4:
            bad += 1
            failures.append({"seed": seed, "shape": [B, H, W], "mask_mismatches": nm, "loss": e_loss, "grad_disp": e_gd,
                             "grad_T": e_gT})
            print(f"seed {seed} {B}x{H}x{W}: mask mismatches {nm}, loss {e_loss:.2e}, gd {e_gd:.2e}, gT {e_gT:.2e}",
                  flush=True)
    if seed % 5 == 4:
        print(f"seed {seed}: failures so far {bad}, worst loss/gd/gT {worst}", flush=True)
summary = {"seeds": n, "cases": 3 * n, "failures": failures, "smoothness_tie_pixels_excluded": n_ties,
           "worst_rel_err": {"loss": worst[0], "grad_disp": worst[1], "grad_T": worst[2]},
           "bars": {"masks": "bit-exact", "loss": 1e-5, "grad": 1e-4}}
print(json.dumps(summary), flush=True)
if len(sys.argv) > 2:
    json.dump(summary, open(sys.argv[2], "w"), indent=1)
sys.exit(1 if bad else 0)
