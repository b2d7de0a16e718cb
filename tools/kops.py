"""Opcode histogram of one kernel in an assembly listing: python tools/kops.py build/meta/photometric.s [kernel-substring]"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else "photo_tile_kernelILb1"
m = re.search(r'^(_ZN\S*' + re.escape(key) + r'\S*):(.*?)\.Lfunc_end', s, re.S | re.M)
ops = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]
c = collections.Counter(o.split()[0] for o in ops)
cls = collections.Counter()
for o, n in c.items():
    k = 'valu' if o.startswith('v_') else 'salu' if o.startswith('s_') else 'lds' if o.startswith('ds_') else 'vmem'
    cls[k] += n
print(m.group(1), len(ops), dict(cls))
print(' '.join(f'{k}:{v}' for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40)))
gl = [o for o in ops if o.startswith('global_load')]
print('global loads', len(gl), 'with scalar base', sum(1 for o in gl if re.search(r's\[\d+:\d+\]', o)))
