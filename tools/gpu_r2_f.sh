#!/bin/bash
mkdir -p gpurun_out/r2f
timeout 1500 python tests/fuzz/net_fuzz.py --cases 500 --seed 21 > gpurun_out/r2f/net_fuzz2.txt 2>&1; echo "net rc=$?"; tail -2 gpurun_out/r2f/net_fuzz2.txt
timeout 1500 python tests/fuzz/path_fuzz.py --cases 300 --seed 21 > gpurun_out/r2f/path_fuzz2.txt 2>&1; echo "path rc=$?"; tail -2 gpurun_out/r2f/path_fuzz2.txt
timeout 1500 python tests/fuzz/conv_fuzz.py --cases 1500 --seed 21 > gpurun_out/r2f/conv_fuzz2.txt 2>&1; echo "conv rc=$?"; tail -2 gpurun_out/r2f/conv_fuzz2.txt
