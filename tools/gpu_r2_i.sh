#!/bin/bash
mkdir -p gpurun_out/r2i
for w in lorenz96 kolmogorov64; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2i/bench_$w.json 2> gpurun_out/r2i/bench_$w.err; tail -c 2500 gpurun_out/r2i/bench_$w.json; tail -3 gpurun_out/r2i/bench_$w.err
done
timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2i/bench_k256.json 2> gpurun_out/r2i/bench_k256.err; tail -c 3500 gpurun_out/r2i/bench_k256.json; tail -3 gpurun_out/r2i/bench_k256.err
# 2-rank launch path on one GPU (gloo): strong + weak
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload kolmogorov64 --per-gpu 4 --steps 2 --warmup 1 --backend gloo --no-profile > gpurun_out/r2i/bench_2rank.json 2> gpurun_out/r2i/bench_2rank.err; tail -c 600 gpurun_out/r2i/bench_2rank.json; tail -3 gpurun_out/r2i/bench_2rank.err
