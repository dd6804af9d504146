#!/bin/bash
mkdir -p gpurun_out/r05c
for m in f16x2 f32; do
  timeout 900 python bench.py --workload kolmogorov64 --steps 10 --warmup 3 --no-cpu-baseline --multiply $m > gpurun_out/r05c/bench_k64_$m.json 2> gpurun_out/r05c/bench_k64_$m.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05c/bench_k64_$m.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$m', 'k64 ms/step', round(d['ms_per_step'],1), 'frac', round(r['frac'],3), 'finite', d['samples_finite'], 'clock', (r.get('clock') or {}).get('probe',{}).get('ghz'))
print('   ', {k:(round(v['ms_per_step'],1), round(v.get('mfma_util',0),3)) for k,v in r['families'].items() if 'ms_per_step' in v})
PY
done
for m in f16x2 f32; do
  timeout 1200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --multiply $m > gpurun_out/r05c/bench_k256_$m.json 2> gpurun_out/r05c/bench_k256_$m.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05c/bench_k256_$m.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$m', 'k256 ms/step', round(d['ms_per_step'],1), 'frac', round(r['frac'],3), 'finite', d['samples_finite'])
print('   ', {k:(round(v['ms_per_step'],1), round(v.get('mfma_util',0),3)) for k,v in r['families'].items() if 'ms_per_step' in v})
PY
done
