#!/bin/bash
mkdir -p gpurun_out/r2m
bash tools/profile_bench.sh r02_kolmogorov256_g1c1 --steps 1 --warmup 1 2>&1 | tail -3
for wl in kolmogorov256 kolmogorov64 lorenz96 lorenz63 qg128; do
  timeout 900 python bench.py --workload $wl --steps 3 --warmup 1 > gpurun_out/r2m/bench_$wl.json 2> gpurun_out/r2m/bench_$wl.err
  python - "$wl" <<'PY'
import json, sys
j=json.loads(open(f'gpurun_out/r2m/bench_{sys.argv[1]}.json').read().strip().split('\n')[-1])
print(sys.argv[1], j['value'], j['ms_per_step'], j.get('roofline', {}).get('frac'), j.get('cpu_baseline', {}).get('value'))
PY
done
python tools/wino4_check.py --skip-check --bench > gpurun_out/r2m/wino4_layer_bench.txt 2>&1
