#!/bin/bash
# round-5 first GPU call: new drop-in / Kolmogorov end-to-end tests, then the default bench line + kolmogorov64 as the clock reference
mkdir -p gpurun_out/r05a
timeout 1500 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_kolmogorov_eval.py tests/test_gpu_net.py -x -q -m gpu -s 2>&1 | grep -v "^\s*[0-9]*%|" | tail -40 > gpurun_out/r05a/tests.log
tail -5 gpurun_out/r05a/tests.log
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err; tail -1 gpurun_out/r05a/bench_default.json | cut -c1-600
timeout 600 python bench.py --workload kolmogorov64 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05a/bench_k64.json 2>/dev/null; tail -1 gpurun_out/r05a/bench_k64.json | cut -c1-300
