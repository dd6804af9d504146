#!/bin/bash
mkdir -p gpurun_out/r2f
timeout 900 python tools/wino4_check.py --cases 80 --variants 0,4 --bench > gpurun_out/r2f/wino4_check.txt 2>&1; echo "rc=$?" >> gpurun_out/r2f/wino4_check.txt
grep -v "^ok" gpurun_out/r2f/wino4_check.txt | tail -40
