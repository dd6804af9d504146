#!/bin/bash
# PMC passes for one conv layer of tools/conv_bench.py (counters collected alone: no sys/hip/hsa trace domains).
# usage: tools/pmc_conv.sh "<--only pattern>" <outdir>
set -u
PAT="$1"; OUT="$2"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python $R/tools/conv_bench.py --only "$PAT" --reps 2 > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'conv_igemm' not in k and 'conv_wino' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print(f'   {c:28s} {v / n:18.1f}  (avg over {n} dispatches)')
PY
