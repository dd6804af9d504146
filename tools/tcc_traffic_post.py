#!/usr/bin/env python3
"""Post-processor of tools/tcc_traffic.sh: per layer of tools/w4_traffic.py (dispatch order; the first launch of each layer dropped), the
L2 request / hit / miss counts and the fabric-side requests next to the algorithmic bytes."""
import csv, glob, json, os, sys
out = sys.argv[1]
K = int(os.environ.get('W4T_K', '3'))
def layer_order(mode):
    for f in sorted(glob.glob(f'{out}/{mode}_*.log')):
        for line in open(f, errors='ignore'):
            line = line.strip()
            if line.startswith('W4T_ORDER '):
                return json.loads(line[len('W4T_ORDER '):])
    return None
def counters(mode, name):
    rows = []
    for f in glob.glob(f'{out}/{mode}_{name}/**/*counter_collection.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = {}
    for r in rows:
        k = r.get('Kernel_Name', '')
        if not ('conv_wino4' in k or 'conv_h2_kernel' in k):
            continue
        per.setdefault(int(r['Dispatch_Id']), {'kernel': k})[r['Counter_Name']] = float(r['Counter_Value'])
    return [per[k] for k in sorted(per)]
res = {}
for mode in ('f32', 'f16x2'):
    order = layer_order(mode)
    if not order:
        print(mode, ': no layer order found'); continue
    merged = None
    for name in ('req', 'rw', 'eard', 'eawr', 'fetch', 'write'):
        c = counters(mode, name)
        if merged is None:
            merged = [dict(kernel=x['kernel']) for x in c]
        if len(c) != len(merged):
            print(f'{mode}/{name}: {len(c)} dispatches vs {len(merged)}'); continue
        for a, b in zip(merged, c):
            a.update({k: v for k, v in b.items() if k != 'kernel'})
    i = 0
    print(f'== multiply {mode}  (per launch; GB = 1e9 B; algorithmic = every operand once)')
    print(f'{"layer":28s} {"kernel":10s} {"alg rd":>7s} {"alg wr":>7s} | {"L2 req M":>9s} {"hit":>6s} | {"EA rd GB":>9s} {"x alg":>6s} {"EA wr GB":>9s} {"x alg":>6s} | {"FETCH_SIZE x2 GB":>16s}')
    for L in order:
        grp = merged[i:i + L['launches']]; i += L['launches']
        grp = grp[1:] or grp
        avg = lambda key: sum(g.get(key, 0.0) for g in grp) / len(grp)
        req, hit, miss = avg('TCC_REQ_sum'), avg('TCC_HIT_sum'), avg('TCC_MISS_sum')
        rd64 = avg('TCC_EA0_RDREQ_sum') - avg('TCC_EA0_RDREQ_32B_sum')
        ea_rd = (rd64 * 64 + avg('TCC_EA0_RDREQ_32B_sum') * 32) / 1e9
        wr64 = avg('TCC_EA0_WRREQ_64B_sum')
        ea_wr = (wr64 * 64 + (avg('TCC_EA0_WRREQ_sum') - wr64) * 32) / 1e9
        kern = 'h2' if 'conv_h2' in grp[0]['kernel'] else 'wino4'
        ar, aw = L['alg_read_bytes'] / 1e9, L['alg_write_bytes'] / 1e9
        row = dict(layer=L['layer'], kernel=kern, alg_read_GB=ar, alg_write_GB=aw, l2_req=req, l2_hit=hit, l2_miss=miss, ea_read_GB=ea_rd, ea_write_GB=ea_wr,
                   fetch_size_x2_GB=2 * avg('FETCH_SIZE') * 1024 / 1e9, write_size_GB=avg('WRITE_SIZE') * 1024 / 1e9)
        res.setdefault(mode, []).append(row)
        print(f'{L["layer"]:28s} {kern:10s} {ar:7.2f} {aw:7.2f} | {req / 1e6:9.1f} {hit / max(req, 1):6.3f} | {ea_rd:9.2f} {ea_rd / ar:6.2f} {ea_wr:9.2f} {ea_wr / aw:6.2f} | {row["fetch_size_x2_GB"]:16.2f}')
json.dump(res, open(f'{out}/tcc_summary.json', 'w'), indent=1)
