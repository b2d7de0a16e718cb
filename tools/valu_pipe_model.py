#!/usr/bin/env python3
"""Vector-pipe cost model of the tile kernel.

The vector pipe of a SIMD does not take the same time for every opcode: measured with
tools/micro/valu_rates.hip (independent chains, four waves per SIMD, profiles/r02_valu_rates.txt),
plain float32 arithmetic (fma, mul, add, sub, fmac, mov, 32-bit integer add) costs about 2.3
cycles per wave-instruction, most other opcodes (compares, conversions, floor, integer min/max,
shifts, selects, DPP, v_div_fixup, integer multiply) about 4.4, 64-bit adds about 4.8 and
transcendentals (rcp, exp) about 8.4.  This script weights the opcode histogram of the kernel's
segments (a -DDW_STAMPS assembly listing: the stamps split prologue, per-scale stages and
epilogue) with those costs and with how often a segment runs per tile (the per-scale stages four
times) and prints the average pipe cycles per vector instruction.  bench.py multiplies that
figure with the SQ_INSTS_VALU counter to state the vector-pipe limit of a launch.

usage: valu_pipe_model.py LISTING.s [--write]   (--write: store the figure in profiles/traffic_latest.json)
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = 2.3
HALF = 4.4
WIDE = 4.8
TRANS = 8.4
FULL_RATE = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_fmaak_f32",
             "v_fmamk_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32"}
TRANSCENDENTAL = {"v_rcp_f32", "v_rcp_iflag_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32"}
WIDE_OPS = {"v_lshl_add_u64", "v_mad_u64_u32", "v_lshlrev_b64", "v_mov_b64"}
PER_TILE = {6: 4, 8: 4, 9: 4, 10: 4, 11: 4}   # segments inside the scale loop (see tools/kseg.py)


def cost(op: str) -> float:
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp"):
        return HALF
    if base in TRANSCENDENTAL:
        return TRANS
    if base in WIDE_OPS:
        return WIDE
    if base in FULL_RATE:
        return FULL
    return HALF


def main():
    text = open(sys.argv[1]).read()
    m = re.search(r"^_ZN2dw17photo_tile_kernelILb1EEEvNS_11PhotoParamsE:(.*?)\.Lfunc_end", text, re.S | re.M)
    segs = [collections.Counter()]
    for ln in m.group(1).split("\n"):
        t = ln.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_memtime":
            segs.append(collections.Counter())
        elif op.startswith("v_"):
            segs[-1][op] += 1
    insts = cycles = 0.0
    for i, seg in enumerate(segs):
        w = PER_TILE.get(i, 1)
        n = sum(seg.values())
        c = sum(k * cost(op) for op, k in seg.items())
        insts += w * n
        cycles += w * c
        if n > 20:
            print(f"segment {i:2d} x{w}: {n:5d} vector instructions, {c:8.0f} pipe cycles ({c / n:.2f} per instruction)")
    per = cycles / insts
    print(f"per tile and wave: {insts:.0f} vector instructions, {cycles:.0f} pipe cycles -> {per:.3f} cycles per instruction")
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "profiles", "traffic_latest.json")
        doc = json.load(open(path))
        doc["valu_pipe_cycles_per_inst"] = per
        doc["valu_pipe_note"] = ("average vector-pipe cycles per vector instruction of photo_tile_kernel<true>: opcode histogram "
                                 "of its segments weighted with the per-opcode costs measured by tools/micro/valu_rates.hip "
                                 "(profiles/r02_valu_rates.txt); made by tools/valu_pipe_model.py")
        json.dump(doc, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
