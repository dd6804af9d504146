#!/bin/bash
mkdir -p gpurun_out/r05d
for nf in 0 3; do
  SDA_NET1D_NF=$nf timeout 300 python bench.py --workload lorenz96 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r05d/l96_nf$nf.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05d/l96_nf$nf.json').read().strip().splitlines()[-1])
print('lorenz96 NF=$nf ms/step', round(d['ms_per_step'],4))
PY
done
timeout 2400 python -m pytest tests/test_gpu_kolmogorov_eval.py tests/test_gpu_dropin.py tests/test_gpu_h2.py tests/test_gpu_configs.py -q -m gpu --durations=20 2>&1 | tail -30
