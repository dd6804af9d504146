#!/bin/bash
# PMC passes (counters alone, one rocprofv3 run per group) for conv_h2_kernel on one plain layer.
# usage: [H2_C=384 H2_H=64 H2_N=120] tools/pmc_h2.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$1"; shift
case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/h2_one.py <<'PY'
import math, os, sys
sys.path.insert(0, os.environ['R'])
import torch
from sda_amd import ops
from sda_amd.engine import launch_conv, planar_source
dev = torch.device('cuda:0')
cin = cout = int(os.environ.get('H2_C', '384')); h = int(os.environ.get('H2_H', '64')); n = int(os.environ.get('H2_N', '120'))
x = torch.randn(n, cin, h, h, device=dev); w = (torch.rand(cout, cin, 3, 3, device=dev) * 2 - 1) / math.sqrt(cin * 9)
pk = ops.PackedConv(w, torch.randn(cout, device=dev)); out = torch.empty(n, cout, h, h, device=dev)
xa = ops.absmax(x, pk.in_amax)
for _ in range(12):
    launch_conv(pk, planar_source(x), out, h, h, circular=True, bias=pk.bias, x_amax=xa)
torch.cuda.synchronize()
PY
export R SDA_MULTIPLY=f16x2
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python /tmp/h2_one.py > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python /tmp/h2_one.py > "$OUT/trace.log" 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(f'{out}/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_h2_kernel' not in r.get('Kernel_Name', ''): continue
        agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
a = {c: agg[c] / cnt[c] for c in agg}
for c in sorted(a):
    print(f'   {c:28s} {a[c]:18.1f}  (avg over {cnt[c]} dispatches)')
ms = None
for f in glob.glob(f'{out}/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_h2_kernel' in r.get('Name', ''):
            ms = float(r['AverageNs']) / 1e6
            print(f'   kernel trace: {r["Calls"]} calls, avg {ms:.3f} ms')
if 'SQ_WAVE_CYCLES' in a:
    wc = a['SQ_WAVE_CYCLES']
    print('   -- per wave-cycle: ' + ', '.join(f'{k}={a[k] / wc:.3f}' for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_WAIT_INST_LDS') if k in a))
if 'GRBM_GUI_ACTIVE' in a and ms:
    print(f"   -- clock under this kernel: GRBM_GUI_ACTIVE / 8 / time = {a['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e-3) / 1e9:.3f} GHz")
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in a:
        print(f"   -- matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * a['GRBM_GUI_ACTIVE'] / 8):.3f}; "
              f"cycles per MFMA = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_INSTS_MFMA']:.1f}")
PY
