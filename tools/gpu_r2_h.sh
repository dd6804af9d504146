#!/bin/bash
mkdir -p gpurun_out/r2h
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
tail -16 gpurun_out/r2h/pytest.log
timeout 600 python bench.py --workload kolmogorov64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2h/bench_k64.json 2> gpurun_out/r2h/bench_k64.err; tail -c 1500 gpurun_out/r2h/bench_k64.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2h/bench_k256.json 2> gpurun_out/r2h/bench_k256.err; tail -c 1500 gpurun_out/r2h/bench_k256.json
