#!/bin/bash
# h2 integration check: unit / net-level tests, layer timing, bench lines in both multiply modes on the same box
mkdir -p gpurun_out/r05c
timeout 1500 python -m pytest tests/test_gpu_h2.py -x -q -m gpu -s 2>&1 | grep -v "^\s*[0-9]*%|" | tail -15 > gpurun_out/r05c/tests_h2.log; tail -6 gpurun_out/r05c/tests_h2.log
SDA_MULTIPLY=f16x2 timeout 600 python tools/h2_check.py --timing-only > gpurun_out/r05c/h2_timing_v6.txt 2>&1; grep -E "plain|SiLU" gpurun_out/r05c/h2_timing_v6.txt | cut -c1-130
for m in f32 f16x2; do
  timeout 900 python bench.py --workload kolmogorov64 --steps 10 --warmup 3 --no-cpu-baseline --multiply $m > gpurun_out/r05c/bench_k64_$m.json 2> gpurun_out/r05c/bench_k64_$m.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05c/bench_k64_$m.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$m', 'k64 ms/step', round(d['ms_per_step'],1), 'frac', round(r['frac'],3), 'dominant', r['kernel'][:30], 'finite', d['samples_finite'], 'clock', (r.get('clock') or {}).get('probe',{}).get('ghz'))
print('   ', {k:(round(v['ms_per_step'],1), round(v.get('mfma_util',0),3)) for k,v in r['families'].items() if 'ms_per_step' in v})
PY
done
for m in f32 f16x2; do
  timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --multiply $m > gpurun_out/r05c/bench_k256_$m.json 2> gpurun_out/r05c/bench_k256_$m.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05c/bench_k256_$m.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$m', 'k256 ms/step', round(d['ms_per_step'],1), 'frac', round(r['frac'],3), 'finite', d['samples_finite'])
print('   ', {k:(round(v['ms_per_step'],1), round(v.get('mfma_util',0),3)) for k,v in r['families'].items() if 'ms_per_step' in v})
PY
done
