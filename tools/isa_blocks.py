#!/usr/bin/env python3
"""Per basic block of one kernel in a hipcc -save-temps .s: instruction class counts (valu / salu / ds / vmem / mfma / wait);
--dump NAME prints one block.    python tools/isa_blocks.py file.s kernel_substring [--min N] [--dump .LBBx_y]"""
import collections, sys
path, pat = sys.argv[1], sys.argv[2]
mn = int(sys.argv[sys.argv.index('--min') + 1]) if '--min' in sys.argv else 20
dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if pat in l and not l.startswith('.') and not l.startswith('\t') and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or lines[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.p2align'):
        continue
    if t.split()[0].endswith(':'):
        blocks.append((name, cur)); cur, name = [], t.split(':')[0]
        continue
    if t.startswith('.'):
        continue
    cur.append(t.split(';')[0].strip())
blocks.append((name, cur))
def cls(i):
    op = i.split()[0]
    if 'mfma' in op: return 'mfma'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('v_'): return 'valu'
    return 'other'
for n, b in blocks:
    if len(b) < mn: continue
    c = collections.Counter(cls(i) for i in b)
    print(f'{n:12s} {len(b):5d}  ' + ' '.join(f'{k}:{v}' for k, v in sorted(c.items())))
    if dump == n:
        print('\n'.join('      ' + i for i in b))
