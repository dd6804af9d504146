#!/bin/bash
mkdir -p gpurun_out/r4d
timeout 1200 python -m pytest tests/test_gpu_fused1d.py -x -q 2>&1 | tail -5
for wl in lorenz63 lorenz96; do
  PROFILE_PMC=0 PROFILE_KERNEL=net1d bash tools/profile_bench.sh r04b_${wl}_g1c1 --workload $wl --steps 200 --warmup 20 2>&1 | head -5
  timeout 600 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r4d/bench_$wl.json 2> /dev/null; cut -c1-200 gpurun_out/r4d/bench_$wl.json
done
