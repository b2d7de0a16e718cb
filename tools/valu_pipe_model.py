#!/usr/bin/env python3
"""Vector-pipe cost model of the tile kernel.

The vector pipe of a SIMD does not take the same time for every instruction.  Measured with
tools/micro/valu_rates.hip (independent chains, every CU busy; profiles/r02_valu_rates.txt), at
two or four waves per SIMD:
  * plain float32 / 32-bit add, sub, logic, mov opcodes in the 32-bit encoding with vector-register,
    inline-constant operands: about 2.55 cycles per wave-instruction;
  * the same in the 64-bit encoding (v_fma_f32, literal operands, neg / abs modifiers): about 3.0;
  * every other opcode (compares, conversions, floor, min / max, shifts, selects, DPP, v_div_fixup,
    integer multiply), AND any opcode of the first two groups with a scalar-register operand:
    about 4.5;  64-bit adds about 4.8;  transcendentals (rcp, exp) about 8.3.
This script weights the instructions of the kernel's segments (a -DDW_STAMPS assembly listing: the
stamps split prologue, per-scale stages and epilogue) with those costs and with how often a
segment runs per tile (the per-scale stages four times) and prints the average pipe cycles per
vector instruction.  bench.py multiplies that figure with the SQ_INSTS_VALU counter to state the
vector-pipe limit of a launch.

usage: valu_pipe_model.py LISTING.s [--write]   (--write: store the figure in profiles/traffic_latest.json)
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL32 = 2.55   # 32-bit encoding, vector-register / inline-constant operands
FULL64 = 3.0    # 64-bit encoding of the same opcodes
HALF = 4.5
WIDE = 4.8
TRANS = 8.3
FULL_RATE = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_fmaak_f32",
             "v_fmamk_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32",
             "v_xor_b32", "v_ashrrev_i32"}
ALWAYS64 = {"v_fma_f32", "v_fmaak_f32", "v_fmamk_f32"}
TRANSCENDENTAL = {"v_rcp_f32", "v_rcp_iflag_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32"}
WIDE_OPS = {"v_lshl_add_u64", "v_mad_u64_u32", "v_lshlrev_b64", "v_mov_b64"}
PER_TILE = {6: 4, 8: 4, 9: 4, 10: 4, 11: 4}   # segments inside the scale loop (see tools/kseg.py)


def cost(line: str) -> float:
    op = line.split()[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp"):
        return HALF
    if base in TRANSCENDENTAL:
        return TRANS
    if base in WIDE_OPS:
        return WIDE
    if base in FULL_RATE:
        srcs = line.split(None, 1)[1].split(",")[1:]
        if any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec)", x) for x in srcs):
            return HALF     # a scalar-register operand
        if base in ALWAYS64 or op.endswith("_e64") or any(re.search(r"0x[0-9a-f]+|\|| -v", x) for x in srcs):
            return FULL64
        return FULL32
    return HALF


def main():
    text = open(sys.argv[1]).read()
    m = re.search(r"^_ZN2dw17photo_tile_kernelILb1EEEvNS_11PhotoParamsE:(.*?)\.Lfunc_end", text, re.S | re.M)
    segs = [[]]
    for ln in m.group(1).split("\n"):
        t = ln.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_memtime":
            segs.append([])
        elif op.startswith("v_"):
            segs[-1].append(t)
    insts = cycles = 0.0
    for i, seg in enumerate(segs):
        w = PER_TILE.get(i, 1)
        n = len(seg)
        c = sum(cost(t) for t in seg)
        insts += w * n
        cycles += w * c
        if n > 20:
            print(f"segment {i:2d} x{w}: {n:5d} vector instructions, {c:8.0f} pipe cycles ({c / n:.2f} per instruction)")
    per = cycles / insts
    print(f"per tile and wave: {insts:.0f} vector instructions, {cycles:.0f} pipe cycles -> {per:.3f} cycles per instruction")
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "profiles", "traffic_latest.json")
        sys.path.insert(0, ROOT)
        from bench import kernel_sources_sha
        sha = kernel_sources_sha()
        doc = json.load(open(path)) if os.path.exists(path) else {}
        if doc.get("sources_sha") != sha:
            # counters of other sources do not belong beside this figure: tools/pmc_traffic.py refills them
            doc = {"kernel": "photo_tile_kernel<true>", "sources_sha": sha, "workloads": {}}
        doc["valu_pipe_cycles_per_inst"] = per
        doc["valu_pipe_note"] = ("average vector-pipe cycles per vector instruction of photo_tile_kernel<true>: opcode histogram "
                                 "of its segments weighted with the per-opcode costs measured by tools/micro/valu_rates.hip "
                                 "(profiles/r02_valu_rates.txt); made by tools/valu_pipe_model.py")
        json.dump(doc, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
