#!/bin/bash
# round-4 second GPU pass: the fused 1-D path
mkdir -p gpurun_out/r4b
timeout 1200 python -m pytest tests/test_gpu_fused1d.py -x -q 2>&1 | tail -30 > gpurun_out/r4b/fused.log; cat gpurun_out/r4b/fused.log
timeout 600 python -m pytest "tests/test_gpu_ops.py::test_wino4_silu_derivative_strongly_negative_preactivation" tests/test_gpu_configs.py::test_config2_full_shard_properties tests/test_gpu_configs.py::test_config4_full_shard_properties -q -s 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r4b/fix.log; cat gpurun_out/r4b/fix.log
for wl in lorenz63 lorenz96; do
  timeout 600 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r4b/bench_$wl.json 2> gpurun_out/r4b/bench_$wl.err; cut -c1-330 gpurun_out/r4b/bench_$wl.json; tail -2 gpurun_out/r4b/bench_$wl.err
done
timeout 900 python bench.py --workload lorenz_eval --no-cpu-baseline > gpurun_out/r4b/bench_lorenz_eval_global_lo.json 2> gpurun_out/r4b/le.err; cut -c1-200 gpurun_out/r4b/bench_lorenz_eval_global_lo.json; tail -2 gpurun_out/r4b/le.err
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r4b/fulltests.log; cat gpurun_out/r4b/fulltests.log
