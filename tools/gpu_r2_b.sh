#!/bin/bash
mkdir -p gpurun_out/r2b
timeout 900 python tools/wino4_check.py --cases 80 --bench > gpurun_out/r2b/wino4_check.txt 2>&1; echo "rc=$?" >> gpurun_out/r2b/wino4_check.txt
tail -60 gpurun_out/r2b/wino4_check.txt
