"""Per-segment instruction classes of the stamped tile kernel: python tools/kseg.py file.s  (built with -DDW_STAMPS)"""
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'^_ZN2dw17photo_tile_kernelILb1EEEvNS_11PhotoParamsE:(.*?)\.Lfunc_end', s, re.S | re.M)
segs = [collections.Counter()]
for ln in m.group(1).split('\n'):
    t = ln.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'): continue
    op = t.split()[0]
    if op == 's_memtime': segs.append(collections.Counter()); continue
    k = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem'
    segs[-1][k] += 1
    for key in ('v_readlane', 'v_writelane', 'v_lshl_add_u64', 's_nop', 's_waitcnt', 'v_mov_b32', 'scratch_'):
        if op.startswith(key): segs[-1][key] += 1
names = {1: 'prologue', 2: 'ystat+identity', 6: 'A', 8: 'B', 10: 'C', 11: 'pose+store', 9: 'planes', 3: 'hdr'}
for i, c in enumerate(segs):
    if sum(c.values()) > 20: print(i, names.get(i, ''), dict(c))
