#!/bin/bash
# PMC passes (counters alone) for the one-wave-per-SIMD Winograd kernel variants on the plain 96->96 @64 layer.
# usage: tools/pmc_w4.sh <outdir> <variant list>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$1"; shift
case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/w4_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['R'])
import torch
from sda_amd import ops
from sda_amd.engine import launch_conv, planar_source
dev = torch.device('cuda:0')
cin = cout = int(os.environ.get('W4_C', '96')); h = int(os.environ.get('W4_H', '64')); n = int(os.environ.get('W4_N', '896'))
x = torch.randn(n, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
pk = ops.PackedConv(w, None); out = torch.empty(n, cout, h, h, device=dev)
for _ in range(3):
    launch_conv(pk, planar_source(x), out, h, h, circular=True)
torch.cuda.synchronize()
PY
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python /tmp/w4_one.py > "$OUT/$name.log" 2>&1
}
for v in "$@"; do
  export SDA_W4_VAR=$v R
  run v${v}_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
  run v${v}_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
  run v${v}_sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT
  run v${v}_grbm GRBM_GUI_ACTIVE GRBM_COUNT
done
cd $R
python - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in sys.argv[2:]:
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(f'{out}/v{v}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_wino4' not in r.get('Kernel_Name', ''): continue
            agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
    print(f'== variant {v}')
    for c in sorted(agg):
        print(f'   {c:28s} {agg[c] / cnt[c]:18.1f}  (avg over {cnt[c]} dispatches)')
    a = {c: agg[c] / cnt[c] for c in agg}
    if 'SQ_WAVE_CYCLES' in a:
        wc = a['SQ_WAVE_CYCLES']
        print('   -- per wave-cycle: ' + ', '.join(f'{k}={a[k] / wc:.3f}' for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_WAIT_INST_LDS') if k in a))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and 'SQ_BUSY_CYCLES' in a:
            print(f"   -- MFMA busy / (4 x SQ busy cycles) = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * a['SQ_BUSY_CYCLES']):.3f}   (raw ratio to SQ_BUSY_CYCLES {a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_BUSY_CYCLES']:.3f})")
PY
