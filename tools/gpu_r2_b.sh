#!/bin/bash
mkdir -p gpurun_out/r2b
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/pytest_all.log 2>&1; tail -6 gpurun_out/r2b/pytest_all.log
for wl in lorenz96 lorenz63 kolmogorov64; do
  timeout 600 python bench.py --workload $wl --steps 100 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import json, sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$wl', j['value'], j['ms_per_step'], j['samples_finite'])"
done
