#!/bin/bash
mkdir -p gpurun_out/r4f
timeout 600 python -m pytest tests/test_gpu_fused1d.py -x -q 2>&1 | tail -3
PROFILE_PMC=0 PROFILE_KERNEL=net1d bash tools/profile_bench.sh r04d_lorenz63_g1c1 --workload lorenz63 --steps 200 --warmup 20 2>&1 | head -4
timeout 600 python tools/lnbwd_fusion_probe.py > gpurun_out/r4f/lnbwd_probe_256.txt 2>&1; cat gpurun_out/r4f/lnbwd_probe_256.txt
timeout 600 python tools/lnbwd_fusion_probe.py --size 64 --n 896 > gpurun_out/r4f/lnbwd_probe_64.txt 2>&1; cat gpurun_out/r4f/lnbwd_probe_64.txt
timeout 600 python tools/w4_quick_bench.py 2>&1 | tail -2 | tee gpurun_out/r4f/w4_quick.txt
