#!/bin/bash
# round-5 closing evidence on ONE box: rocprofv3 kernel stats + PMC passes of the default bench command (fp32 route and the opt-in f16x2
# route), then the driver's bench command line.  Post-process here: python tools/profile_post.py r05_kolmogorov256_g1c1 ;
# python tools/profile_post.py r05_kolmogorov256_g1c1_f16x2 --kernel conv_h2_kernel
mkdir -p gpurun_out/r05_final
bash tools/profile_bench.sh r05_kolmogorov256_g1c1 --steps 1 --warmup 1 --second-line 0 > gpurun_out/r05_final/prof_f32.log 2>&1
PROFILE_KERNEL=conv_h2_kernel bash tools/profile_bench.sh r05_kolmogorov256_g1c1_f16x2 --steps 1 --warmup 1 --multiply f16x2 > gpurun_out/r05_final/prof_f16x2.log 2>&1
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_final/bench_driver_cmd.json 2> gpurun_out/r05_final/bench_driver_cmd.err
tail -c 900 gpurun_out/r05_final/bench_driver_cmd.json
