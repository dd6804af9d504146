#!/bin/bash
mkdir -p gpurun_out/r2l
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2l/pytest.log 2>&1; tail -3 gpurun_out/r2l/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r2l/bench_k256.json 2> gpurun_out/r2l/bench_k256.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2l/bench_k256.json').read().strip().split('\n')[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'])
for k,v in j['roofline']['families'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
