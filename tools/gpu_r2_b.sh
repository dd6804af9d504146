#!/bin/bash
for wl in lorenz96 lorenz63; do
  timeout 600 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import json, sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$wl', j['value'], j['ms_per_step'], j['samples_finite'])"
done
