#!/bin/bash
mkdir -p gpurun_out/r2k
timeout 900 python tools/wino4_check.py --cases 120 --bench > gpurun_out/r2k/check3.txt 2>&1; echo "rc=$?" >> gpurun_out/r2k/check3.txt
grep -v "^ok" gpurun_out/r2k/check3.txt | tail -30
