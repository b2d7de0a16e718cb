#!/usr/bin/env python3
"""Instruction mix and register use of the compute_loss strip kernels (assembly listing from -save-temps)."""
import collections, re, sys
t = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/compute_loss-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for mode in ("Li0ELb0Ef", "Li1ELb0Ef", "Li1ELb0Ed", "Li0ELb1Ef"):
    m = re.search(r"^_ZN2dw25compute_loss_strip_kernelI%sEEvNS_8ClParamsE:(.*?)\.Lfunc_end" % mode, t, re.S | re.M)
    ops = [l.split()[0] for l in m.group(1).split("\n") if re.match(r"\s+(v_|s_|ds_|global_|buffer_)", l)]
    c = collections.Counter(re.sub(r"_e(32|64)$", "", o) for o in ops)
    nv = sum(n for o, n in c.items() if o.startswith("v_"))
    meta = re.search(r"\.name:\s+_ZN2dw25compute_loss_strip_kernelI%sEEvNS_8ClParamsE.*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)" % mode, t, re.S)
    print(mode, "instructions", len(ops), "vector", nv, "per step ~%.0f" % (nv / 7), "vgprs", meta.group(1), "spills", meta.group(2))
    print("  ", sorted(c.items(), key=lambda kv: -kv[1])[:16])
