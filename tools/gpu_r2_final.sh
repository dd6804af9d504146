#!/bin/bash
mkdir -p gpurun_out/r2final
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2final/smoke.log 2>&1; tail -2 gpurun_out/r2final/smoke.log
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r2final/pytest.log 2>&1; tail -2 gpurun_out/r2final/pytest.log
timeout 900 python bench.py > gpurun_out/r2final/bench_default.json 2> gpurun_out/r2final/bench_default.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2final/bench_default.json').read().strip().split('\n')[-1])
print({k: j[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','dtype')})
print(j['config']['workload'], j['roofline']['frac'], j['roofline']['achieved'], j['cpu_baseline'])
PY
for wl in lorenz96 lorenz63; do
  timeout 900 python bench.py --workload $wl --steps 50 --warmup 5 > gpurun_out/r2final/bench_$wl.json 2> /dev/null
  python - "$wl" <<'PY'
import json, sys
j=json.loads(open(f'gpurun_out/r2final/bench_{sys.argv[1]}.json').read().strip().split('\n')[-1])
print(sys.argv[1], j['value'], j['ms_per_step'], j.get('cpu_baseline', {}).get('value'))
PY
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload kolmogorov64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2final/bench_2rank.json 2> gpurun_out/r2final/bench_2rank.err; tail -1 gpurun_out/r2final/bench_2rank.json | cut -c1-300; tail -2 gpurun_out/r2final/bench_2rank.err | cut -c1-300
