#!/bin/bash
# round-4 final evidence on one box: headline bench, its rocprofv3 passes (kernel trace + PMC), the other workloads' lines
mkdir -p gpurun_out/r4z
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r4z/kolmogorov256_g1c1_bench.json 2> gpurun_out/r4z/k256.err; cut -c1-260 gpurun_out/r4z/kolmogorov256_g1c1_bench.json
mkdir -p gpurun_out/profiles_out
keep() { cp profiles/$1_* gpurun_out/profiles_out/ 2>/dev/null; rm -rf gpurun_out/prof_$1; }   # (the raw rocprofv3 output exceeds what gpurun carries back)
bash tools/profile_bench.sh r04_kolmogorov256_g1c1 --steps 1 --warmup 1 2>&1 | tail -5; keep r04_kolmogorov256_g1c1
for net in global local; do for fr in lo hi; do
  timeout 900 python bench.py --workload lorenz_eval --lorenz-net $net --lorenz-freq $fr --cpu-seconds 8 > gpurun_out/r4z/lorenz_eval_${net}_${fr}_bench.json 2> /dev/null
done; done
for wl in lorenz63 lorenz96; do
  timeout 900 python bench.py --workload $wl --steps 200 --warmup 20 > gpurun_out/r4z/${wl}_g1c1_bench.json 2> /dev/null
  PROFILE_PMC=0 PROFILE_KERNEL=net1d bash tools/profile_bench.sh r04_${wl}_g1c1 --workload $wl --steps 200 --warmup 20 > /dev/null 2>&1; keep r04_${wl}_g1c1
done
for wl in kolmogorov64 qg128; do
  timeout 900 python bench.py --workload $wl > gpurun_out/r4z/${wl}_g1c1_bench.json 2> /dev/null
  PROFILE_PMC=0 bash tools/profile_bench.sh r04_${wl}_g1c1 --workload $wl --steps 1 --warmup 1 > /dev/null 2>&1; keep r04_${wl}_g1c1
done
python tools/mlp_bench.py > gpurun_out/r4z/mlp_bench.txt 2>&1
ls gpurun_out/r4z
du -sh gpurun_out
