#!/usr/bin/env python3
"""Disparity and pose gradients of one make_sequence case against the oracle in float32 and float64:
python tools/grad_disp_f64.py B H W SEED"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_estimation_amd.synth import make_sequence
from tests.test_photometric_gpu import _oracle, _hip, smoothness_ties
B, H, W, seed = (int(v) for v in sys.argv[1:5])
seq = make_sequence(B, H, W, seed=seed)
_, _, _, gd32, gT32 = _oracle(seq)
_, _, _, gd64, gT64 = _oracle(seq.to("cpu", torch.float64))
_, _, hgd, hgT = _hip(seq)
for s in range(4):
    keep = (~smoothness_ties(seq.disp[s])).double()
    sc = gd64[s].abs().max()
    eh = ((hgd[s].double() - gd64[s]).abs() * keep) / sc
    eo = ((gd32[s].double() - gd64[s]).abs() * keep) / sc
    e32 = ((hgd[s] - gd32[s]).abs() * keep.float()) / gd32[s].abs().max()
    i = int(e32.argmax())
    print(f"scale {s}: hip-f64 {float(eh.max()):.2e} f32-f64 {float(eo.max()):.2e} hip-f32 {float(e32.max()):.2e}"
          f" | at worst hip-f32 pixel: hip-f64 {float(eh.flatten()[i]):.2e} f32-f64 {float(eo.flatten()[i]):.2e}")
sc = gT64[:, :, :3].abs().max()
print("T: hip-f64 %.2e f32-f64 %.2e" % (float((hgT[:, :, :3].double() - gT64[:, :, :3]).abs().max() / sc),
                                       float((gT32[:, :, :3].double() - gT64[:, :, :3]).abs().max() / sc)))
