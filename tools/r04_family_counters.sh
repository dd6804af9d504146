#!/bin/bash
# counters of the non-dominant convolution families inside a configs[3] step (same passes as tools/profile_bench.sh, one summary per kernel)
mkdir -p gpurun_out/profiles_out
bash tools/profile_bench.sh r04_k256_fam --steps 1 --warmup 1 > /dev/null 2>&1
for k in 'conv_wino4_kernel<false, true, false, 1, 0, 1>:w4zp_up' 'conv_wino4_kernel<false, false, false, 0, 0, 2>:w4zp_pooled' 'conv_igemm_ws_kernel:direct' 'conv_par4_kernel:par4' 'conv_few_kernel:few'; do
  python tools/profile_post.py r04_kolmogorov256_${k##*:} --kernel "${k%%:*}" --src gpurun_out/prof_r04_k256_fam > /dev/null 2>&1
done
rm -f profiles/r04_kolmogorov256_{w4zp_up,w4zp_pooled,direct,par4,few}_kernel_stats.csv profiles/r04_kolmogorov256_{w4zp_up,w4zp_pooled,direct,par4,few}_bench_under_rocprof.json profiles/r04_k256_fam_*
cp profiles/r04_kolmogorov256_{w4zp_up,w4zp_pooled,direct,par4,few}_traffic.json gpurun_out/profiles_out/; rm -rf gpurun_out/prof_r04_k256_fam; ls gpurun_out/profiles_out
