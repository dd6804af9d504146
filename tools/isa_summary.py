#!/usr/bin/env python3
"""Summarise the gfx950 assembly of one kernel (hipcc -save-temps .s): per basic block with MFMAs, the instruction
histogram, the MFMA / filler interleave pattern and compiler-inserted waits -- what one reads before spending GPU time.

    python tools/isa_summary.py file.s kernel_substring [--dump N]
"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
dump = int(sys.argv[sys.argv.index('--dump') + 1]) if '--dump' in sys.argv else -1
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if pat in l and not l.startswith('.') and not l.startswith('\t') and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or lines[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.p2align'):
        continue
    if t.endswith(':') or (t.split()[0].endswith(':')):
        blocks.append((name, cur)); cur, name = [], t.split(':')[0]
        continue
    if t.startswith('.'):
        continue
    cur.append(t.split(';')[0].strip())
blocks.append((name, cur))
tot = collections.Counter()
for n, b in blocks:
    for ins in b:
        tot[ins.split()[0]] += 1
print('kernel total instructions:', sum(tot.values()), ' mfma:', sum(v for k, v in tot.items() if 'mfma' in k),
      ' scratch:', sum(v for k, v in tot.items() if 'scratch' in k), ' accvgpr moves:', sum(v for k, v in tot.items() if 'accvgpr' in k))
k = 0
for n, b in blocks:
    nm = sum('mfma' in i for i in b)
    if nm < 8:
        continue
    h = collections.Counter(i.split()[0] for i in b)
    print(f'--- block {n}: {len(b)} instr, {nm} mfma')
    print('   ', ', '.join(f'{a}:{c}' for a, c in h.most_common(24)))
    # interleave pattern: number of non-MFMA instructions between consecutive MFMAs
    gaps, g = [], 0
    for i in b:
        if 'mfma' in i:
            gaps.append(g); g = 0
        else:
            g += 1
    print('    fillers before each mfma:', ' '.join(map(str, gaps)), '| tail', g)
    waits = [i for i in b if i.startswith('s_waitcnt')]
    print('    waits:', collections.Counter(waits).most_common(12))
    if k == dump:
        print('\n'.join('      ' + i for i in b))
    k += 1
