#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite) kernel trace: per-kernel calls / total / average / share
(the rocpd `top_kernels` view reports durations in microseconds).

    python tools/rocpd_summary.py results.db [--csv out.csv]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, total_calls, total_duration, average, percentage from top_kernels '
                  'order by total_duration desc').fetchall()
out = ['name,calls,total_ms,avg_us,percent']
for name, calls, tot, avg, pct in rows:
    short = name if len(name) < 110 else name[:107] + '...'
    out.append(f'"{short}",{calls},{tot / 1e3:.3f},{avg:.1f},{pct:.2f}')
text = '\n'.join(out)
print(text)
if '--csv' in sys.argv:
    open(sys.argv[sys.argv.index('--csv') + 1], 'w').write(text + '\n')
