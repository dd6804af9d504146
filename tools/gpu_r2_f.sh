#!/bin/bash
mkdir -p gpurun_out/r2f
timeout 1500 python tests/fuzz/conv_fuzz.py --cases 2500 --seed 7 > gpurun_out/r2f/conv_fuzz.txt 2>&1; echo "conv rc=$?"; tail -3 gpurun_out/r2f/conv_fuzz.txt
timeout 1500 python tests/fuzz/conv_fuzz.py --cases 400 --large --seed 3 > gpurun_out/r2f/conv_fuzz_large.txt 2>&1; echo "conv large rc=$?"; tail -3 gpurun_out/r2f/conv_fuzz_large.txt
timeout 1500 python tests/fuzz/net_fuzz.py --cases 400 --seed 5 > gpurun_out/r2f/net_fuzz.txt 2>&1; echo "net rc=$?"; tail -3 gpurun_out/r2f/net_fuzz.txt
timeout 1500 python tests/fuzz/path_fuzz.py --cases 300 --seed 5 > gpurun_out/r2f/path_fuzz.txt 2>&1; echo "path rc=$?"; tail -3 gpurun_out/r2f/path_fuzz.txt
timeout 900 python tests/fuzz/ops_fuzz.py --cases 300 --seed 5 > gpurun_out/r2f/ops_fuzz.txt 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/r2f/ops_fuzz.txt
