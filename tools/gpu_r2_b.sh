#!/bin/bash
mkdir -p gpurun_out/r2final
for wl in lorenz96 lorenz63; do
  timeout 900 python bench.py --workload $wl --steps 100 --warmup 5 > gpurun_out/r2final/bench_$wl.json 2> /dev/null
  python - "$wl" <<'PY'
import json, sys
j=json.loads(open(f'gpurun_out/r2final/bench_{sys.argv[1]}.json').read().strip().split('\n')[-1])
print(sys.argv[1], j['value'], j['ms_per_step'], j.get('cpu_baseline', {}).get('value'))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2final/l96 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload lorenz96 --steps 20 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $(ls gpurun_out/r2final/l96/*.db | head -1) | head -6
