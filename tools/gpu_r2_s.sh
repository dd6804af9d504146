#!/bin/bash
mkdir -p gpurun_out/r2s
timeout 1500 python tests/fuzz/conv_fuzz.py --cases 800 --seed 11 > gpurun_out/r2s/conv_fuzz.txt 2>&1; echo "conv rc=$?"; tail -2 gpurun_out/r2s/conv_fuzz.txt
timeout 1500 python tests/fuzz/net_fuzz.py --cases 150 --seed 9 > gpurun_out/r2s/net_fuzz.txt 2>&1; echo "net rc=$?"; tail -2 gpurun_out/r2s/net_fuzz.txt
for wl in lorenz96 lorenz63; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2s/bench_$wl.json 2> gpurun_out/r2s/bench_$wl.err
  python - "$wl" <<'PY'
import json, sys
j=json.loads(open(f'gpurun_out/r2s/bench_{sys.argv[1]}.json').read().strip().split('\n')[-1])
print(sys.argv[1], j['value'], j['ms_per_step'])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2s/l96 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload lorenz96 --steps 20 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $(ls gpurun_out/r2s/l96/*.db | head -1) | head -8
