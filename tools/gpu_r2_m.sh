#!/bin/bash
mkdir -p gpurun_out/r2m
for wl in kolmogorov256 kolmogorov64 lorenz96 lorenz63 qg128; do
  timeout 900 python bench.py --workload $wl --steps 3 --warmup 1 > gpurun_out/r2m/bench_$wl.json 2> gpurun_out/r2m/bench_$wl.err
  python - "$wl" <<'PY'
import json, sys
j=json.loads(open(f'gpurun_out/r2m/bench_{sys.argv[1]}.json').read().strip().split('\n')[-1])
print(sys.argv[1], j['value'], j['ms_per_step'], j.get('roofline', {}).get('frac'), j.get('cpu_baseline', {}).get('value'))
PY
done
for wl in lorenz96 lorenz63; do
  timeout 900 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r2m/bench_${wl}_200.json 2> /dev/null
  python - "$wl" <<'PY'
import json, sys
j=json.loads(open(f'gpurun_out/r2m/bench_{sys.argv[1]}_200.json').read().strip().split('\n')[-1])
print(sys.argv[1], '200 steps', j['value'], j['ms_per_step'])
PY
done
