#!/bin/bash
# What a round-end check runs on the GPU box (via gpurun): smoke, the GPU test-suite, the default bench line, the other workloads.
#   /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/gpu_check.sh'
mkdir -p gpurun_out/check
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/check/bench_default.json 2> gpurun_out/check/bench_default.err; tail -1 gpurun_out/check/bench_default.json | cut -c1-400
for wl in kolmogorov64 qg128 lorenz96 lorenz63; do
  timeout 900 python bench.py --workload $wl --steps 20 --warmup 3 > gpurun_out/check/bench_$wl.json 2> /dev/null; tail -1 gpurun_out/check/bench_$wl.json | cut -c1-200
done
# multi-rank launch path on a 1-GPU box: two ranks over gloo, both on cuda:0 (the driver runs the RCCL curve on an 8-GPU node)
timeout 900 python bench.py --gpus 2 --backend gloo --workload kolmogorov64 --steps 4 --warmup 1 \
  > gpurun_out/check/bench_2rank_gloo_kolmogorov64.json 2> gpurun_out/check/bench_2rank_gloo.err    # (no torchrun: bench.py launches its ranks)
tail -1 gpurun_out/check/bench_2rank_gloo_kolmogorov64.json | cut -c1-300
