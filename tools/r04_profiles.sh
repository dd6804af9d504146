#!/bin/bash
# round-4 evidence: bench lines + rocprofv3 kernel traces for every workload (PMC passes for configs[3] only)
mkdir -p gpurun_out/r4p
for wl in lorenz63 lorenz96; do
  timeout 900 python bench.py --workload $wl --steps 200 --warmup 20 > gpurun_out/r4p/${wl}_g1c1_bench.json 2> gpurun_out/r4p/${wl}.err
  PROFILE_PMC=0 PROFILE_KERNEL=net1d bash tools/profile_bench.sh r04_${wl}_g1c1 --workload $wl --steps 200 --warmup 20 > /dev/null 2>&1
done
for wl in kolmogorov64 qg128; do
  timeout 1200 python bench.py --workload $wl --steps 20 --warmup 3 > gpurun_out/r4p/${wl}_g1c1_bench.json 2> gpurun_out/r4p/${wl}.err
  PROFILE_PMC=0 bash tools/profile_bench.sh r04_${wl}_g1c1 --workload $wl --steps 4 --warmup 1 > /dev/null 2>&1
done
for net in global local; do for fr in lo hi; do
  timeout 900 python bench.py --workload lorenz_eval --lorenz-net $net --lorenz-freq $fr --cpu-seconds 8 > gpurun_out/r4p/lorenz_eval_${net}_${fr}_bench.json 2> gpurun_out/r4p/le_${net}_$fr.err
done; done
timeout 900 python bench.py --gpus 2 --backend gloo --workload kolmogorov64 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r4p/kolmogorov64_2rank_gloo_selflaunch_bench.json 2> gpurun_out/r4p/2rank.err
ls -la gpurun_out/r4p | head -30
