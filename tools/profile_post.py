#!/usr/bin/env python3
"""Post-process the rocprofv3 passes of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into profiles/<tag>_traffic.json.
Runs on the GPU box at the end of profile_bench.sh, or here on the merged gpurun_out/ (profiles/ written on the box does not
travel back):   python tools/profile_post.py gpurun_out/prof_<tag> <tag> [profiles]"""
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
prof = sys.argv[3] if len(sys.argv) > 3 else 'profiles'
def avg(which):
    tot, n = 0.0, 0
    for f in glob.glob(f'{out}/{which}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' in r.get('Kernel_Name', '') or 'conv_wino' in r.get('Kernel_Name', ''):
                tot += float(r['Counter_Value']); n += 1
    return (tot / n if n else None), n
def avg_k(which, counter, kern):
    tot, n = 0.0, 0
    for f in glob.glob(f'{out}/{which}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r.get('Kernel_Name', '') and r.get('Counter_Name') == counter:
                tot += float(r['Counter_Value']); n += 1
    return (tot / n if n else None), n
# dominant kernel alone: MFMA instruction count and busy cycles per launch, GPU-active cycles per launch (GRBM, summed over the
# 8 XCDs): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8)
dom = {}
for c in ('SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU'):
    dom[c], dom['launches_sq_pass'] = avg_k('sq', c, 'conv_wino4')
dom['GRBM_GUI_ACTIVE'], dom['launches_grbm_pass'] = avg_k('grbm', 'GRBM_GUI_ACTIVE', 'conv_wino4')
if dom.get('SQ_VALU_MFMA_BUSY_CYCLES') and dom.get('GRBM_GUI_ACTIVE'):
    dom['mfma_util_from_counters'] = dom['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (dom['GRBM_GUI_ACTIVE'] / 8.0)
    dom['issued_flop_per_launch'] = dom['SQ_INSTS_MFMA'] * 2 * 16 * 16 * 4 * 1.0
fw, nfw = avg_k('fetch', 'FETCH_SIZE', 'conv_wino4'); ww, nww = avg_k('write', 'WRITE_SIZE', 'conv_wino4')
f, nf = avg('fetch'); w, nw = avg('write')
res = {'kernel': 'conv_wino4_kernel (dominant) ; all conv kernels in *_all fields', 'launches_fetch_pass': nfw, 'launches_write_pass': nww,
       'dominant_kernel_counters_per_launch': dom,
       'FETCH_SIZE_KiB_per_launch_raw_all_conv': f, 'WRITE_SIZE_KiB_per_launch_raw_all_conv': w,
       'FETCH_SIZE_KiB_per_launch_raw': fw, 'WRITE_SIZE_KiB_per_launch_raw': ww,
       'hbm_bytes_per_launch': None if fw is None or ww is None else (2 * fw + ww) * 1024,
       'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE reports 0.500 of the true bytes for 16 B/lane and 4 B/lane streams and for conv_wino4 halo rows alike (128-B lines), WRITE_SIZE 1.00-1.10 (calibration on known byte counts: tools/fetch_calib.hip, profiles/r03_w4_traffic.json)'}
json.dump(res, open(f'{prof}/{tag}_traffic.json', 'w'), indent=1)
print(res)
