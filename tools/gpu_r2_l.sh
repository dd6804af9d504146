#!/bin/bash
mkdir -p gpurun_out/r2l
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2l/pytest.log 2>&1; tail -5 gpurun_out/r2l/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r2l/bench_k256.json 2> gpurun_out/r2l/bench_k256.err; cat gpurun_out/r2l/bench_k256.json | cut -c1-2500
timeout 600 python bench.py --workload kolmogorov64 --steps 5 --warmup 2 > gpurun_out/r2l/bench_k64.json 2> gpurun_out/r2l/bench_k64.err; cat gpurun_out/r2l/bench_k64.json | cut -c1-600
