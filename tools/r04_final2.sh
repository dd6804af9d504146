#!/bin/bash
# round-4 closing evidence on one box (after the ln_bwd change): GPU suite, the driver's headline command, its rocprofv3 passes
mkdir -p gpurun_out/r4f gpurun_out/profiles_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4f/kolmogorov256_g1c1_bench.json 2> gpurun_out/r4f/k256.err; cut -c1-260 gpurun_out/r4f/kolmogorov256_g1c1_bench.json
bash tools/profile_bench.sh r04_kolmogorov256_g1c1 --steps 1 --warmup 1 2>&1 | tail -5
# the same counter passes, averaged over the LayerNorm kernels of the 96-channel level (HBM-bound: bytes per launch against 4 c hw 4 B / image)
python tools/profile_post.py r04_kolmogorov256_lnbwd96 --kernel "ln_bwd_quad_kernel<8, 12, 1>" --src gpurun_out/prof_r04_kolmogorov256_g1c1 > /dev/null 2>&1
python tools/profile_post.py r04_kolmogorov256_lnstats96 --kernel "ln_stats_quad_kernel<8, 12>" --src gpurun_out/prof_r04_kolmogorov256_g1c1 > /dev/null 2>&1
rm -f profiles/r04_kolmogorov256_ln*_kernel_stats.csv profiles/r04_kolmogorov256_ln*_bench_under_rocprof.json
cp profiles/r04_kolmogorov256_g1c1_* profiles/r04_kolmogorov256_ln*_traffic.json gpurun_out/profiles_out/ 2>/dev/null; rm -rf gpurun_out/prof_r04_kolmogorov256_g1c1
timeout 600 python bench.py --workload kolmogorov64 > gpurun_out/r4f/kolmogorov64_g1c1_bench.json 2> /dev/null; cut -c1-200 gpurun_out/r4f/kolmogorov64_g1c1_bench.json
du -sh gpurun_out
