#!/usr/bin/env python3
"""THE post-processor of tools/profile_bench.sh's rocprofv3 passes (gpurun_out/prof_<tag>/) -- the only writer of
profiles/<tag>_kernel_stats.csv, profiles/<tag>_traffic.json and profiles/<tag>_bench_under_rocprof.json.  Runs here on the merged
gpurun_out/ (profiles/ written on the GPU box does not travel back) or on the box at the end of profile_bench.sh:

    python tools/profile_post.py <tag> [--kernel SUBSTR] [--src gpurun_out/prof_<tag>] [--dst profiles]

--kernel selects the DOMINANT kernel the per-launch counters are averaged over (default: conv_wino4_kernel's full 16-position
variants, template argument ZP = 0; "net1d" for the Lorenz workloads).  Every figure in profiles/README.md and DESIGN 5.3 that
comes from counters is one field of <tag>_traffic.json.
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument('tag')
ap.add_argument('--kernel', default='conv_wino4_kernel')
ap.add_argument('--src')
ap.add_argument('--dst')
ap.add_argument('--rederive', action='store_true',
                help='no raw passes at hand: recompute the DERIVED fields of an existing profiles/<tag>_traffic.json from the raw per-launch '
                     'counters stored in it (used once, for r05_..._f16x2: its issued flops had been priced at the fp32 MFMA shape)')
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = args.src or os.path.join(root, 'gpurun_out', f'prof_{args.tag}')
dst = args.dst or os.path.join(root, 'profiles')


if args.rederive:
    fn = os.path.join(args.dst or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles'), f'{args.tag}_traffic.json')
    doc = json.load(open(fn))
    kern = doc.get('kernel', '')
    dm = doc['dominant_kernel_counters_per_launch']
    fpm = 2 * 32 * 32 * 16 if 'conv_h2' in kern else 2 * 16 * 16 * 4
    dm['flop_per_mfma_instruction'] = fpm
    dm['issued_flop_per_launch'] = dm['SQ_INSTS_MFMA'] * fpm * 1.0
    doc['rederived'] = 'issued_flop_per_launch = SQ_INSTS_MFMA x flop per MFMA of the kernel\'s instruction (tools/profile_post.py --rederive)'
    json.dump(doc, open(fn, 'w'), indent=1)
    print(fn, dm['issued_flop_per_launch'])
    raise SystemExit(0)


def is_dominant(name: str) -> bool:
    if args.kernel not in name:
        return False
    if args.kernel == 'conv_wino4_kernel':              # full variants only: <MOD, LN, SILU, EPM, VAR, ZP> with ZP == 0
        m = re.search(r'<([^>]*)>', name)
        if m:
            targs = [a.strip() for a in m.group(1).split(',')]
            return len(targs) < 6 or targs[5] == '0'
    return True


# ---- 1. kernel trace (--kernel-trace --stats): per-kernel calls / total / average
dbs = glob.glob(os.path.join(src, 'trace', '*.db'))
stats = []
if dbs:
    stats = sqlite3.connect(dbs[0]).execute('select name, total_calls, total_duration, average, percentage from top_kernels '
                                            'order by total_duration desc').fetchall()
    with open(os.path.join(dst, f'{args.tag}_kernel_stats.csv'), 'w') as f:
        f.write('name,calls,total_ms,avg_us,percent\n')
        for name, calls, tot, avg, pct in stats:
            short = name if len(name) < 110 else name[:107] + '...'
            f.write(f'"{short}",{calls},{tot / 1e3:.3f},{avg:.1f},{pct:.2f}\n')
if os.path.exists(os.path.join(src, 'bench_trace.json')):
    shutil.copy(os.path.join(src, 'bench_trace.json'), os.path.join(dst, f'{args.tag}_bench_under_rocprof.json'))


# ---- 2. counters (separate --pmc passes): per-launch averages over the dominant kernel's launches
def avg(which: str, counter: str, pred):
    tot, n = 0.0, 0
    for fn in glob.glob(f'{src}/{which}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get('Counter_Name') == counter and pred(r.get('Kernel_Name', '')):
                tot += float(r['Counter_Value'])
                n += 1
    return (tot / n if n else None), n


dom = {}
for c in ('SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
          'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU'):
    dom[c], dom['launches_sq_pass'] = avg('sq', c, is_dominant)
dom['GRBM_GUI_ACTIVE'], dom['launches_grbm_pass'] = avg('grbm', 'GRBM_GUI_ACTIVE', is_dominant)
if dom.get('SQ_VALU_MFMA_BUSY_CYCLES') and dom.get('GRBM_GUI_ACTIVE'):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
    dom['mfma_util_from_counters'] = dom['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (dom['GRBM_GUI_ACTIVE'] / 8.0)
    # flop per MFMA instruction of the dominant kernel: v_mfma_f32_16x16x4_f32 (conv_wino4 and every fp32 kernel here but the 32x32x2 ones,
    # which issue the same 2 048) vs v_mfma_f32_32x32x16_f16 (conv_h2: 2 x 32 x 32 x 16 = 32 768) -- VERDICT r5 weak 5: this was 2 048 for all
    flop_per_mfma = 2 * 32 * 32 * 16 if 'conv_h2' in args.kernel else 2 * 16 * 16 * 4
    dom['flop_per_mfma_instruction'] = flop_per_mfma
    dom['issued_flop_per_launch'] = dom['SQ_INSTS_MFMA'] * flop_per_mfma * 1.0
    dom['valu_per_mfma'] = (dom['SQ_INSTS_VALU'] or 0.0) / dom['SQ_INSTS_MFMA'] if dom.get('SQ_INSTS_MFMA') else None
dur = [(calls, tot) for name, calls, tot, a, p in stats if is_dominant(name)]
if dur:
    dom['kernel_trace_launches'] = sum(c for c, _ in dur)
    dom['kernel_trace_avg_ms'] = sum(t for _, t in dur) / 1e3 / max(1, dom['kernel_trace_launches'])
    if dom.get('GRBM_GUI_ACTIVE'):
        dom['sustained_clock_GHz'] = dom['GRBM_GUI_ACTIVE'] / 8.0 / (dom['kernel_trace_avg_ms'] * 1e-3) / 1e9
fw, nfw = avg('fetch', 'FETCH_SIZE', is_dominant)
ww, nww = avg('write', 'WRITE_SIZE', is_dominant)
allconv = lambda k: 'conv_' in k
fa, _ = avg('fetch', 'FETCH_SIZE', allconv)
wa, _ = avg('write', 'WRITE_SIZE', allconv)
res = {'tag': args.tag, 'kernel': f'{args.kernel} (dominant' + (', full 16-position variants: ZP = 0' if args.kernel == 'conv_wino4_kernel' else '') + ')',
       'written_by': 'tools/profile_post.py (the only writer of this file)',
       'launches_fetch_pass': nfw, 'launches_write_pass': nww,
       'dominant_kernel_counters_per_launch': dom,
       'FETCH_SIZE_KiB_per_launch_raw': fw, 'WRITE_SIZE_KiB_per_launch_raw': ww,
       'hbm_bytes_per_launch': None if fw is None or ww is None else (2 * fw + ww) * 1024,
       'FETCH_SIZE_KiB_per_launch_raw_all_conv_kernels': fa, 'WRITE_SIZE_KiB_per_launch_raw_all_conv_kernels': wa,
       'correction': 'bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: counters are KiB, separate --pmc passes, counters alone; on gfx950 '
                     'FETCH_SIZE reports 0.500 of the true bytes for 16 B/lane and 4 B/lane streams and for conv_wino4 halo rows alike '
                     '(128-B lines), WRITE_SIZE 1.00-1.10 (calibration on known byte counts in the kernel\'s own access shapes: '
                     'tools/fetch_calib.hip, profiles/r03_w4_traffic.txt)'}
if any(glob.glob(f'{src}/{w}/**/*counter_collection.csv', recursive=True) for w in ('sq', 'grbm', 'fetch', 'write')):
    json.dump(res, open(os.path.join(dst, f'{args.tag}_traffic.json'), 'w'), indent=1)       # (kernel-trace-only runs: no counter file)
for name, calls, tot, a, p in stats[:8]:
    print(f'{p:6.2f} %  {calls:6d} x {a / 1e3:9.4f} ms  {name[:100]}')
print(json.dumps(res, indent=1))
