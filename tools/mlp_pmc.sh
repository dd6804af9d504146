#!/bin/bash
# PMC counters of the whole-MLP kernels (separate passes; kernel trace + counters only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mlp_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o p -- python $R/tools/mlp_bench.py > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- python $R/tools/mlp_bench.py > /dev/null 2> $OUT/grbm.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/lds -o p -- python $R/tools/mlp_bench.py > /dev/null 2> $OUT/lds.err
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/mlp_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mlp_' in k:
            key = (k.split('(')[0][-18:], r['Counter_Name'])
            acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f'{k:20s} {c:28s} {v / n:16.0f}  ({n} launches)')
PY
