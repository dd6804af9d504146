#!/bin/bash
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "fused_1d or unet1d or lorenz or hipgraph" > gpurun_out/r2b/pytest1.log 2>&1; tail -5 gpurun_out/r2b/pytest1.log
SDA_BLOCK1D_TP=64 timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "fused_1d" > gpurun_out/r2b/pytest2.log 2>&1; tail -3 gpurun_out/r2b/pytest2.log
for tp in 32 64; do for wl in lorenz96 lorenz63; do
  SDA_BLOCK1D_TP=$tp timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 2>gpurun_out/r2b/bench_$wl.err | python -c "
import json, sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('tp$tp', '$wl', j['value'], j['ms_per_step'])"
done; done
