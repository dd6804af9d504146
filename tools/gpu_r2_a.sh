#!/bin/bash
# round-2 call A: full GPU suite (new config parity tests included) + baseline layer benches
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -30 gpurun_out/r2a/pytest.log
timeout 300 python tools/conv_bench.py --n 896 --size 64 > gpurun_out/r2a/convbench_64.txt 2>&1
timeout 300 python tools/conv_bench.py --n 60 --size 256 > gpurun_out/r2a/convbench_256.txt 2>&1
cat gpurun_out/r2a/convbench_64.txt gpurun_out/r2a/convbench_256.txt
