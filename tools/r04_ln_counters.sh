#!/bin/bash
# HBM-traffic counters of the LayerNorm kernels inside a configs[3] step (same passes as tools/profile_bench.sh, summarised per LN kernel)
mkdir -p gpurun_out/profiles_out
bash tools/profile_bench.sh r04_k256_ln --steps 1 --warmup 1 > /dev/null 2>&1
for k in 'ln_bwd_quad_kernel<8, 12, 1, 1>:lnbwd96' 'ln_bwd_quad_kernel<8, 24, 1, 1>:lnbwd192' 'ln_bwd_quad_kernel<16, 24, 1, 1>:lnbwd384' 'ln_stats_quad_kernel<8, 12, 1>:lnstats96' 'ln_stats_quad_kernel<8, 24, 1>:lnstats192' 'ln_stats_quad_kernel<8, 12, 4>:lnstats384'; do
  python tools/profile_post.py r04_kolmogorov256_${k##*:} --kernel "${k%%:*}" --src gpurun_out/prof_r04_k256_ln > /dev/null 2>&1
done
rm -f profiles/r04_kolmogorov256_ln*_kernel_stats.csv profiles/r04_kolmogorov256_ln*_bench_under_rocprof.json profiles/r04_k256_ln_*
cp profiles/r04_kolmogorov256_ln*_traffic.json gpurun_out/profiles_out/; rm -rf gpurun_out/prof_r04_k256_ln; ls gpurun_out/profiles_out
