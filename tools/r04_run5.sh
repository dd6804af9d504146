#!/bin/bash
mkdir -p gpurun_out/r4e
timeout 1500 python -m pytest tests/test_gpu_fused1d.py tests/test_gpu_net.py tests/test_gpu_configs.py::test_config1_lorenz96_guided_eager_and_graph tests/test_gpu_configs.py::test_config0_lorenz63_256_steps_end_to_end tests/test_gpu_lorenz_eval.py -x -q 2>&1 | tail -8
for wl in lorenz63 lorenz96; do
  PROFILE_PMC=0 PROFILE_KERNEL=net1d bash tools/profile_bench.sh r04c_${wl}_g1c1 --workload $wl --steps 200 --warmup 20 2>&1 | head -5
  timeout 600 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r4e/bench_$wl.json 2> /dev/null; cut -c1-200 gpurun_out/r4e/bench_$wl.json
done
timeout 900 python bench.py --workload lorenz_eval --no-cpu-baseline > gpurun_out/r4e/bench_lorenz_eval_global_lo.json 2> gpurun_out/r4e/le.err; cut -c1-200 gpurun_out/r4e/bench_lorenz_eval_global_lo.json
