#!/usr/bin/env python3
"""Post-process tools/traffic.sh: per layer, FETCH_SIZE / WRITE_SIZE per conv_wino4 launch against the algorithmic bytes, for
the contiguous (walk0) and the XCD-interleaved (walk1) tile walk, with the calibration factors of tools/fetch_calib.hip.

    python tools/w4_traffic_post.py <outdir> [json-out]
"""
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]


def counter_rows(d, kernel_pat):
    rows = []
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(kernel_pat, r.get('Kernel_Name', '')):
                rows.append((int(r.get('Dispatch_Id', 0)), r['Kernel_Name'], r['Counter_Name'], float(r['Counter_Value'])))
    rows.sort()
    return rows


res = {'units': 'FETCH_SIZE / WRITE_SIZE are reported in KiB; bytes = value * 1024'}
# ---- calibration
exp = {}
log = os.path.join(out, 'calib_fetch.log')
if os.path.exists(log):
    for line in open(log):
        t = line.split()
        if t[:1] == ['expect']:
            exp[t[1]] = t[2:]
    cal = {}
    for cname, d in (('FETCH_SIZE', 'calib_fetch'), ('WRITE_SIZE', 'calib_write')):
        per = {}
        for _, k, c, v in counter_rows(os.path.join(out, d), r'stream16|stream4|halo_read|store8'):
            if c != cname:
                continue
            key = 'halo_iso' if 'halo_read<0>' in k else 'halo_dense' if 'halo_read<1>' in k else \
                [n for n in ('stream16', 'stream4', 'store8') if k.startswith(n)][0]
            per.setdefault(key, []).append(v * 1024)
        cal[cname] = {k: sum(v) / len(v) for k, v in per.items()}
    res['calibration_bytes_reported'] = cal
    f = cal.get('FETCH_SIZE', {})
    region = float(exp['stream16'][0])
    iso = {k: float(v) for k, v in zip(exp['halo_iso'][0::2], exp['halo_iso'][1::2])}
    res['calibration'] = {
        'stream16_reported_over_true': f.get('stream16', 0) / region,
        'stream4_reported_over_true': f.get('stream4', 0) / region,
        'halo_iso_reported_over_useful': f.get('halo_iso', 0) / iso['useful'],
        'halo_iso_reported_over_64B_segments': f.get('halo_iso', 0) / iso['seg64'],
        'halo_iso_reported_over_128B_lines': f.get('halo_iso', 0) / iso['line128'],
        'halo_dense_reported_over_plane_bytes': f.get('halo_dense', 0) / region,
        'store8_WRITE_reported_over_true': cal.get('WRITE_SIZE', {}).get('store8', 0) / region,
        'stream_reads_WRITE_reported': cal.get('WRITE_SIZE', {}).get('stream16', 0),
    }
# ---- per-layer traffic
layers = {}
for walk in (0, 1):
    for cname, tag in (('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')):
        d = os.path.join(out, f'w4t_walk{walk}_{tag}')
        logf = d + '.log'
        if not os.path.exists(logf):
            continue
        order = None
        for line in open(logf):
            if line.startswith('W4T_ORDER '):
                order = json.loads(line[len('W4T_ORDER '):])
        if order is None:
            continue
        vals = [v * 1024 for _, k, c, v in counter_rows(d, 'conv_wino4_kernel') if c == cname]
        pos = 0
        for o in order:
            chunk = vals[pos:pos + o['launches']]
            pos += o['launches']
            if not chunk:
                continue
            use = chunk[1:] if len(chunk) > 1 else chunk        # (the first launch of a layer also pulls the freshly packed weights)
            e = layers.setdefault(o['layer'], dict(alg_read_bytes=o['alg_read_bytes'], alg_write_bytes=o['alg_write_bytes']))
            e[f'walk{walk}_{cname}_bytes'] = sum(use) / len(use)
res['layers'] = layers
corr = None
if 'calibration' in res and res['calibration']['halo_dense_reported_over_plane_bytes'] > 0:
    corr = 1.0 / res['calibration']['halo_dense_reported_over_plane_bytes']
    res['fetch_correction_used'] = {'factor': corr, 'why': 'true bytes / reported bytes of the dense halo read (the kernel\'s own access shape, every byte fetched once)'}
print(f"{'layer':28s} {'alg read':>9s} {'walk0 raw':>10s} {'x alg':>6s} {'walk1 raw':>10s} {'x alg':>6s} | {'alg write':>9s} {'walk1 W':>9s} {'x alg':>6s}   (GB per launch" + (f'; corrected fetch = raw x {corr:.2f})' if corr else ')'))
for name, e in layers.items():
    a = e['alg_read_bytes']
    f0, f1 = e.get('walk0_FETCH_SIZE_bytes'), e.get('walk1_FETCH_SIZE_bytes')
    w1 = e.get('walk1_WRITE_SIZE_bytes') or e.get('walk0_WRITE_SIZE_bytes')
    fmt = lambda v, ref: (f'{v / 1e9:10.3f} {v / ref:6.2f}' if v else f'{"-":>10s} {"-":>6s}')
    print(f"{name:28s} {a / 1e9:9.3f} {fmt(f0, a)} {fmt(f1, a)} | {e['alg_write_bytes'] / 1e9:9.3f} " + (f"{w1 / 1e9:9.3f} {w1 / e['alg_write_bytes']:6.2f}" if w1 else ''))
    if corr:
        for k in ('walk0', 'walk1'):
            if e.get(f'{k}_FETCH_SIZE_bytes'):
                e[f'{k}_fetch_corrected_over_alg'] = e[f'{k}_FETCH_SIZE_bytes'] * corr / a
if 'calibration' in res:
    print(json.dumps(res['calibration'], indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], 'w'), indent=1)
