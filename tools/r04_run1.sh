#!/bin/bash
# round-4 first GPU pass: new tests, self-launching 2-rank bench, Lorenz baselines, lorenz_eval
mkdir -p gpurun_out/r4a
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_lorenz_eval.py tests/test_gpu_net.py::test_nonlinear_masked_coupled_observations_without_autograd \
  "tests/test_gpu_ops.py::test_wino4_silu_derivative_strongly_negative_preactivation" tests/test_gpu_configs.py::test_config2_full_shard_properties \
  tests/test_gpu_configs.py::test_config4_full_shard_properties -x -q -s 2>&1 | tail -25 > gpurun_out/r4a/newtests.log
cat gpurun_out/r4a/newtests.log
timeout 600 python bench.py --gpus 2 --backend gloo --workload kolmogorov64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4a/bench_2rank_selflaunch.json 2> gpurun_out/r4a/bench_2rank.err
tail -c 600 gpurun_out/r4a/bench_2rank_selflaunch.json; tail -3 gpurun_out/r4a/bench_2rank.err
for wl in lorenz63 lorenz96; do
  timeout 600 python bench.py --workload $wl --steps 200 --warmup 20 > gpurun_out/r4a/bench_$wl.json 2> gpurun_out/r4a/bench_$wl.err; cut -c1-300 gpurun_out/r4a/bench_$wl.json
done
for net in global local; do for fr in lo hi; do
  timeout 900 python bench.py --workload lorenz_eval --lorenz-net $net --lorenz-freq $fr --cpu-seconds 8 > gpurun_out/r4a/bench_lorenz_eval_${net}_$fr.json 2> gpurun_out/r4a/bench_lorenz_eval_${net}_$fr.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4a/bench_lorenz_eval_${net}_$fr.json').read().strip().splitlines()[-1])
    print('${net} ${fr}', 'six-run s', d['six_run_wallclock_s'], 'steps/s', d['value'], {k:round(v['ms_per_step'],3) for k,v in d['per_C'].items()}, d['config']['hipgraph_note'], 'cpu', d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print('${net} ${fr} failed', e); print(open('gpurun_out/r4a/bench_lorenz_eval_${net}_$fr.err').read()[-1500:])
PY
done; done
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r4a/fulltests.log; cat gpurun_out/r4a/fulltests.log
