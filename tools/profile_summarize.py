#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_bench.sh (merged back under gpurun_out/prof_<tag>/) into the small
committed evidence files:  profiles/<tag>_kernel_stats.csv, profiles/<tag>_traffic.json, profiles/<tag>_bench_under_rocprof.json

    python tools/profile_summarize.py <tag>
"""
import csv
import glob
import json
import os
import shutil
import sqlite3
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', f'prof_{tag}')
dst = os.path.join(root, 'profiles')

db = glob.glob(os.path.join(src, 'trace', '*.db'))[0]
rows = sqlite3.connect(db).execute('select name, total_calls, total_duration, average, percentage from top_kernels '
                                   'order by total_duration desc').fetchall()
with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w') as f:
    f.write('name,calls,total_ms,avg_us,percent\n')
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        f.write(f'"{short}",{calls},{tot / 1e3:.3f},{avg:.1f},{pct:.2f}\n')


def avg(which):
    tot, n = 0.0, 0
    for fn in glob.glob(f'{src}/{which}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r.get('Kernel_Name', '')
            if 'conv_igemm' in k or 'conv_wino' in k:
                tot += float(r['Counter_Value'])
                n += 1
    return (tot / n if n else None), n


fv, nf = avg('fetch')
wv, nw = avg('write')
res = {'kernels': 'conv_wino_kernel + conv_igemm*', 'launches_fetch_pass': nf, 'launches_write_pass': nw,
       'FETCH_SIZE_KiB_per_launch_raw': fv, 'WRITE_SIZE_KiB_per_launch_raw': wv,
       'hbm_bytes_per_launch': None if fv is None or wv is None else (2 * fv + wv) * 1024,
       'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: counters are KiB; on gfx950 FETCH_SIZE reports half of a wide '
                     'coalesced read stream (MI355X_MICROARCH.md, HBM section); separate --pmc passes, counters alone'}
json.dump(res, open(os.path.join(dst, f'{tag}_traffic.json'), 'w'), indent=1)
shutil.copy(os.path.join(src, 'bench_trace.json'), os.path.join(dst, f'{tag}_bench_under_rocprof.json'))
print(open(os.path.join(dst, f'{tag}_kernel_stats.csv')).read().split('\n')[:6])
print(res)
