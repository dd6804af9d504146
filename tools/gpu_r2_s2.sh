#!/bin/bash
for v in 16384 4096 1024 64; do
for wl in lorenz96 lorenz63; do
  SDA_LN_SMALL_PIXELS=$v timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import json, sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', '$wl', j['value'], j['ms_per_step'])"
done; done
